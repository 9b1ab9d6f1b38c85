#!/bin/bash
# round 6, final validation: smoke(), the whole GPU suite, then the round's profile pass (tools/profile_round.sh r6) on the final binary
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6final7
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -4 $O/smoke.log
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
bash tools/profile_round.sh r6 > $R/gpurun_out/r6_profile.log 2>&1
tail -3 $R/gpurun_out/r6_profile.log
