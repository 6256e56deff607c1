#!/bin/bash
# The round's profile passes on the GPU box (run through gpurun from the repo root):
#   scripts/profile_round.sh <tag>      -> gpurun_out/<tag>/{stats,stats_n6,fetch,write}/..., bench*.json
# rocprofv3 kernel stats of the bench command (n = 3 headline, n = 6 shard), the two PMC passes (one counter per
# pass, --kernel-trace only) and the default bench line.  Summaries are copied into profiles/ by hand afterwards.
set -e -o pipefail
tag=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
B="python3 $R/bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 20 --warmup 3 --no-cpu-baseline --no-aux > $O/bench_under_rocprof.json 2> $O/stats.err
echo "stats n3 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_n6 -- $B --segments 6 --directions 256 --steps 24 --warmup 3 --no-cpu-baseline --no-aux > $O/bench_n6_under_rocprof.json 2> $O/stats_n6.err
echo "stats n6 done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 5 --warmup 2 --no-cpu-baseline --no-aux > $O/bench_fetch.json 2> $O/fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 5 --warmup 2 --no-cpu-baseline --no-aux > $O/bench_write.json 2> $O/write.err
echo "write done"
cd $R
python3 bench.py > $O/bench.json 2> $O/bench.err
echo "bench done"
tail -c 600 $O/bench.json
