#!/bin/bash
# Small-frame regime on the GPU box: the small_frames leg of bench.py, its kernel trace and its HIP API trace.
#   usage: tools/gpu_small_frames.sh [tag=r06_small] [calls=200]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-r06_small}; CALLS=${2:-200}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python bench.py --small-frames $CALLS > $O/small_frames.json 2> $O/small_frames.err; python3 -c "
import json; d = json.load(open('$O/small_frames.json'))['small_frames']
for k, v in d.items():
    if isinstance(v, dict) and 'plain' in v:
        for m in ('plain', 'group_cull'):
            if m in v: e = v[m]; print('%-40s %-10s call %8.1f us (min %8.1f)  kernel %8.1f us  overhead %6.1f us  %8.1f Msamples/s  device-resident %8.1f us  grid %d' % (k, m, e['call_us_median'], e['call_us_min'], e['kernel_us_median'], e['overhead_us'], e['Msamples_per_s'], e['device_resident_us_median'], e['grid_blocks']))
"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --small-frames 50 > $O/trace.log 2>&1
for f in $O/trace/*kernel_stats.csv; do echo "== $f"; cat $f; done
rocprofv3 --hip-trace --stats --output-format csv -d $O/hip -o h -- python $R/bench.py --small-frames 50 > $O/hip.log 2>&1
for f in $O/hip/*hip_api_stats.csv $O/hip/*hip_stats.csv; do [ -f $f ] && { echo "== $f"; head -30 $f; }; done
