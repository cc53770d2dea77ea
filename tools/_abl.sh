TAG=merged timeout 120 python tools/mb_staged.py 2>&1 | grep -v amdgpu.ids
CHECK=0 TAG=merged-g0 FP8Q_STAGED_GRID=0 timeout 100 python tools/mb_staged.py 2>&1 | grep -v amdgpu.ids
CHECK=0 TAG=merged-g1024 FP8Q_STAGED_GRID=1024 timeout 100 python tools/mb_staged.py 2>&1 | grep -v amdgpu.ids
