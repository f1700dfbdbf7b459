set -u
echo "== split"; python tools/exp_hybrid.py 64 2>&1 | grep -v amdgpu.ids
echo "== f32"; AMAR_PAIR_MFMA=f32 python tools/exp_hybrid.py 64 2>&1 | grep -v amdgpu.ids
