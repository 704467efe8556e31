#!/bin/bash
# round 4, call AS: whole GPU suite + smoke + full bench on the final tree
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04_as
mkdir -p $OUT
export PYTHONPATH=$R:$R/sgl-kernel-xpu_amd/python
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -c 400 $OUT/bench.json; tail -2 $OUT/bench.err
