#!/bin/bash
# Round-4 supporting evidence on ONE box (after tools/round_profiles.sh): tools/round4_extra.sh   (outputs under gpurun_out/r04x/)
#   per-kernel time of one training step (T = 1 B = 216, T = 3 B = 36, bf16x3 B = 216), same-box A/B of the new head engines and of the
#   overlapped optimizer, per-operation micro-benchmarks of the head (old engine vs new)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r04x; mkdir -p $OUT; cd $R
bash tools/prof_step.sh 216 r04_b216 > /dev/null 2>&1; cp gpurun_out/step_r04_b216/train_step_kernels.txt $OUT/train_step_kernels_b216.txt
bash tools/prof_step.sh 36 r04_t3 --temporal 3 --classes 13 > /dev/null 2>&1; cp gpurun_out/step_r04_t3/train_step_kernels.txt $OUT/train_step_kernels_t3_b36.txt
bash tools/prof_step.sh 216 r04_x3 --precision bf16x3 > /dev/null 2>&1; cp gpurun_out/step_r04_x3/train_step_kernels.txt $OUT/train_step_kernels_x3_b216.txt
IG_CONV8=0 IG_WGRAD8_CONV=0 bash tools/prof_step.sh 216 r04_b216_old > /dev/null 2>&1; cp gpurun_out/step_r04_b216_old/train_step_kernels.txt $OUT/train_step_kernels_b216_round1_head.txt
bash tools/ab_step.sh 2 "IG_CONV8=0 IG_WGRAD8_CONV=0" "IG_CONV8=1 IG_WGRAD8_CONV=1" 2>&1 | grep -v amdgpu > $OUT/ab_head_engines_b216.log
bash tools/ab_step.sh 2 "IG_CONV8=0 IG_WGRAD8_CONV=0" "IG_CONV8=1 IG_WGRAD8_CONV=1" --temporal 3 --classes 13 --batch 36 2>&1 | grep -v amdgpu > $OUT/ab_head_engines_t3_b36.log
bash tools/ab_step.sh 1 "IG_CONV8=0 IG_WGRAD8_CONV=0" "IG_CONV8=1 IG_WGRAD8_CONV=1" --temporal 3 --classes 13 --batch 8 2>&1 | grep -v amdgpu > $OUT/ab_head_engines_t3_b8.log
bash tools/ab_step.sh 2 "IG_ADAMW_OVERLAP=0" "IG_ADAMW_OVERLAP=1" 2>&1 | grep -v amdgpu > $OUT/ab_adamw_overlap_b216.log
timeout 400 python tools/head_bench.py --batch 216 2>&1 | grep -v amdgpu.ids > $OUT/head_bench_b216.log
timeout 400 python tools/head_bench.py --batch 36 --temporal 3 2>&1 | grep -v amdgpu.ids > $OUT/head_bench_t3_b36.log
timeout 400 python tools/head_bench.py --batch 216 --split 2>&1 | grep -v amdgpu.ids > $OUT/head_bench_b216_bf16x3.log
head -3 $OUT/*.log | cut -c1-200
