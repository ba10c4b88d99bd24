#!/bin/bash
# round 4: the per-sample engine after the lastArg fix: drop-in parity + real-time factors
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04am
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_host.py -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 900 python tools/dropin_rates.py --out $O/dropin.md > $O/dropin.log 2>&1
grep -v amdgpu $O/dropin.md | cut -c1-200 | head -40
