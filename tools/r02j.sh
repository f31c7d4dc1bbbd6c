set -eu
O=gpurun_out/r02j; mkdir -p $O
python tools/step_ramp.py > $O/ramp3.txt 2>&1; grep -v amdgpu.ids $O/ramp3.txt
