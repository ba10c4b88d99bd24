#!/bin/bash
# round 6: the sample bank's counters again on the final kernel (ring rows)
cd $GRAFT_REPO_ROOT
ONLY="sample_bank" bash tools/profile_r06.sh r06 2>&1 | tail -3
