#!/bin/bash
# GPU box: the evidence files of a round in one call.  tools/final_profiles.sh <tag>  ->  gpurun_out/<tag>/
#   gpu_tests.txt        python -m pytest tests -m gpu
#   bench.json           python bench.py (the driver's command)
#   bench_kernel_stats.txt   rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline` (all blocks), top rows
#   path_b_frame.txt     one steady-state MVSNet forward, kernels grouped by name
#   path_a_kernel_stats.txt / path_a_frame.txt   the same for RobustMVD.forward (tools/run_path_a.py)
tag=${1:-final}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; python tools/print_bench.py $O/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fp_b /tmp/fp_a
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_b -o b -- python3 $R/bench.py --no-cpu-baseline --no-path-a --steps 20 > $O/prof_b.log 2>&1
python3 $R/tools/kernel_stats_table.py $(find /tmp/fp_b -name "*kernel_stats.csv" | head -1) 40 > $O/bench_kernel_stats.txt
python3 $R/tools/frame_kernels.py $(find /tmp/fp_b -name "*kernel_trace.csv" | head -1) warp_variance_tile > $O/path_b_frame.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_a -o a -- python3 $R/tools/run_path_a.py 2 12 > $O/prof_a.log 2>&1
python3 $R/tools/kernel_stats_table.py $(find /tmp/fp_a -name "*kernel_stats.csv" | head -1) 40 > $O/path_a_kernel_stats.txt
python3 $R/tools/frame_kernels.py $(find /tmp/fp_a -name "*kernel_trace.csv" | head -1) sweep_corr > $O/path_a_frame.txt
head -12 $O/path_a_frame.txt
