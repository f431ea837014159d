#!/bin/bash
# evidence for the OPT-IN split-bf16 configuration of Stage I (frozen ViT forward + backward products): bench lines (f32, forward only, forward + backward)
# and a rocprofv3 kernel-stats pass
R=$GRAFT_REPO_ROOT; cd $R
common="--stage 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads"
ACT_TEACHER_BF16X3=0 timeout 300 python bench.py $common > gpurun_out/r05_stage1_f32_same_session.json 2> /dev/null
ACT_TEACHER_BF16X3=1 ACT_TEACHER_BF16X3_BWD=0 timeout 300 python bench.py $common > gpurun_out/r05_stage1_split_bf16_opt_in_fwd_only.json 2> /dev/null
ACT_TEACHER_BF16X3=1 timeout 300 python bench.py $common > gpurun_out/r05_stage1_split_bf16_opt_in.json 2> gpurun_out/r05_stage1_split_bf16_opt_in.err
cd /tmp; export TMPDIR=/tmp
ACT_TEACHER_BF16X3=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s1x3 -- python $R/bench.py --stage 1 --steps 12 --warmup 4 --no-cpu-baseline --no-instrument --no-other-workloads > $R/gpurun_out/prof_s1x3.log 2>&1
cd $R
find gpurun_out/prof_s1x3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_s1_x3optin_kernel_stats.csv
rm -rf gpurun_out/prof_s1x3
for f in r05_stage1_f32_same_session r05_stage1_split_bf16_opt_in_fwd_only r05_stage1_split_bf16_opt_in; do python -c "
import json,sys; d=json.loads(open('gpurun_out/$f.json').read()); print('$f', d['metric'], round(d['ms_per_step'],2), round(d['value'],1))"; done
head -12 gpurun_out/r05_s1_x3optin_kernel_stats.csv | cut -c1-150
