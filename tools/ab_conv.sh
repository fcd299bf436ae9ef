# A/B of conv kernel variants (tools/kbench.py conv): usage: bash tools/ab_conv.sh [variant ...]
cd /tmp && export TMPDIR=/tmp
for v in main "$@"; do
  lib=/root/repo/fudanocr_amd/libfocr_hip.so; [ $v != main ] && lib=/root/repo/fudanocr_amd/libfocr_hip_$v.so
  echo "== $v"; FOCR_LIB=$lib python /root/repo/tools/kbench.py conv 2>&1 | grep -E "srb|linear 128->128|up 3x3|crnn 3x3 256" | cut -c1-130
done
