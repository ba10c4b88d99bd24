# round 5: rocprofv3 evidence of config 5's one-launch form (kernel trace + stats, WRITE_SIZE / FETCH_SIZE passes, bench line)
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r05
ONLY="config5" bash tools/profile_r05.sh r05 2>&1 | tail -5
du -sh gpurun_out/prof_r05
