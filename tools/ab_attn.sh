# A/B of attention kernel variants under a kernel trace: usage: bash tools/ab_attn.sh [variant ...]
cd /tmp && export TMPDIR=/tmp
for v in main "$@"; do
  lib=/root/repo/fudanocr_amd/libfocr_hip.so; [ $v != main ] && lib=/root/repo/fudanocr_amd/libfocr_hip_$v.so
  FOCR_LIB=$lib rocprofv3 --kernel-trace -d /tmp/ab_$v -o p -- python /root/repo/tools/kbench.py attn > /tmp/ab_$v.log 2>&1
  echo "== $v"; python /root/repo/tools/rocpd_stats.py $(find /tmp/ab_$v -name "*.db" | head -1) /tmp/ab_$v.csv 2>/dev/null; grep attn /tmp/ab_$v.csv | awk -F, '{printf "%-60s n=%4d avg=%8.1f min=%8.1f\n", substr($1,1,60), $2, $4, $5}'
done
