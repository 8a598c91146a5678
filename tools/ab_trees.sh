#!/bin/bash
# Same-box A/B of two trees on the GPU box: a bench line (default: C2, no extras; AB_ARGS="--config C3" etc.) alternating
# old / new, three times.
#   here:        git worktree add _ab_old <commit> && make -C _ab_old/lidar_odometry_demo_amd/csrc && cp oracle/*.so _ab_old/oracle/
#   on the box:  gpurun -- 'bash tools/ab_trees.sh'          (_ab_old/ is git-ignored and travels with the snapshot)
#   afterwards:  git worktree remove --force _ab_old
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/ab
for i in 1 2 3; do
  for t in old new; do
    if [ $t = old ]; then D=$R/_ab_old; else D=$R; fi
    (cd $D && timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras ${AB_ARGS:-} 2>/dev/null) > $R/gpurun_out/ab/ab_$t.json
    python3 -c "
import json; d=json.load(open('$R/gpurun_out/ab/ab_$t.json')); r=d['roofline']; print('$t', round(d['ms_per_step'],4), 'k_match', round(r['avg_launch_us'],2), 'k_lm', round(d['roofline_kernels'][1]['avg_us'],2))"
  done
done
