python -m pytest tests/test_models_gpu.py -x -q -k "partitioned_runner" > gpurun_out/r4_t6.log 2>&1; tail -3 gpurun_out/r4_t6.log
O=gpurun_out/r4_rank_of_n_c.txt; : > $O
for ph in 0 1; do
  echo "# AMAR_PART_PHASES=$ph  python tools/exp_rank_of_n.py 64 8 4 2 (EXP_WIRE=1, rank 0)" >> $O
  AMAR_PART_PHASES=$ph EXP_SINGLE_MS=1.023 EXP_WIRE=1 EXP_RANKS=0 python tools/exp_rank_of_n.py 64 8 4 2 >> $O 2>&1
done
for ph in 0 1; do
  echo "# AMAR_PART_PHASES=$ph  python tools/exp_rank_of_n.py 256 8 (EXP_WIRE=1, rank 0)" >> $O
  AMAR_PART_PHASES=$ph EXP_SINGLE_MS=5.24 EXP_WIRE=1 EXP_RANKS=0 python tools/exp_rank_of_n.py 256 8 >> $O 2>&1
done
for ph in 0 1; do
  echo "# AMAR_PART_PHASES=$ph  EXP_MODEL=HybridBertGCN python tools/exp_rank_of_n.py 64 8 (EXP_WIRE=1, rank 0)" >> $O
  AMAR_PART_PHASES=$ph EXP_SINGLE_MS=5.47 EXP_MODEL=HybridBertGCN EXP_WIRE=1 EXP_RANKS=0 python tools/exp_rank_of_n.py 64 8 >> $O 2>&1
done
grep -v amdgpu.ids $O | cut -c1-250
