#!/bin/bash
# on the GPU box: where the resolve's first wave takes over (kTailLive / kTailQueries of orbt_kernels.hip) -- variants built here as
# build_ub/t<live>/liborbslamm_hip.so (-DORBT_TAIL_LIVE=.. -DORBT_TAIL_QUERIES=..), picked up through LD_LIBRARY_PATH
R=${GRAFT_REPO_ROOT:-/root/repo}
X=$R/examples/multi_robot
J() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', {k:d[k] for k in d if k in ('frames_per_s','ms_median','ms_p99','checksum')})"; }
for rep in 1 2 3; do for v in default ${VARIANTS:-t256 t1024 t2048 t4096}; do
  if [ $v = default ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$R/build_ub/$v; fi
  $X --mode track --robots 1 --frames 1500 --interval 0 --json | J "$v one robot      "
  $X --mode full --robots 1 --frames 1000 --interval 0 --json | J "$v one robot, full"
done; done
