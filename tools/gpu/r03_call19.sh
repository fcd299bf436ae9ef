#!/bin/bash
# halo kernel: stagger co-resident blocks so one block's staging/stores overlap the other's MFMA phase (sweep).
# Run from a build in which the delay and the block-id bit were runtime parameters (env below); the kernel now keeps one
# compile-time point: hipcc -DH3_STAGGER=<units of 1024 clocks> tools/ubench/conv_ubench.cpp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for sh in 5 0 6; do for st in 0 4 8 12 16 24 32; do
  echo -n "shift $sh stagger $st: "
  FOCR_H3_STAGGER=$st FOCR_H3_STAGGER_SHIFT=$sh timeout 60 build/conv_ubench 128 "srb 3x3" | grep srb | sed 's/.*halo x3/halo x3/'
done; done
} > gpurun_out/c19_stagger.log 2>&1
cat gpurun_out/c19_stagger.log
