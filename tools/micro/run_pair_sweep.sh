set -u
for wg in 32 64 128; do echo "== AMAR_SCATTER_WG=$wg"; AMAR_SCATTER_WG=$wg AMAR_PAIR_PROJ=0 python tools/exp_pair_parts.py 64 48 2>&1 | grep "window"; done
