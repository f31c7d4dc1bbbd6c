cd $GRAFT_REPO_ROOT
bash tools/profile.sh r05f > gpurun_out/prof_r05f.log 2>&1; echo "profile 65536x1 done"
python tools/update_traffic.py gpurun_out/prof_r05f 65536x1 r05_65536x1_summary.txt
ENVS=4096 bash tools/profile.sh r05f_4096 > gpurun_out/prof_r05f_4096.log 2>&1; echo "profile 4096x1 done"
python tools/update_traffic.py gpurun_out/prof_r05f_4096 4096x1 r05_4096x1_summary.txt
ENVS=16384 AGENTS=2 bash tools/profile.sh r05f_16384x2 > gpurun_out/prof_r05f_16384x2.log 2>&1; echo "profile 16384x2 done"
python tools/update_traffic.py gpurun_out/prof_r05f_16384x2 16384x2 r05_16384x2_summary.txt
cp profiles/traffic.json gpurun_out/traffic_r05.json
bash tools/profile_default.sh > gpurun_out/r05_default_trace.txt 2>&1; cat gpurun_out/r05_default_trace.txt
