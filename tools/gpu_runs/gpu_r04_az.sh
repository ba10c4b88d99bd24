#!/bin/bash
# round 4: blocks that are only 8-byte aligned fall back to the 8-byte streams (every bank kernel with a pair-row form)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_edges.py -q -x 2>&1 | tail -15 | cut -c1-250
