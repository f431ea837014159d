cd $GRAFT_REPO_ROOT
echo "== XCD placement of the NT kernel (teacher shapes), TF by ACT_GEMM_XCD_ROWS"
for r in 0 1 2 4; do echo "-- xcd_rows=$r"; ACT_GEMM_XCD_ROWS=$r python benchmarks/gemm_bench.py vit. t8. dgcnn.l5 2>&1 | grep -E "^(vit|t8|dgcnn)" | cut -c1-110; done
echo "== Stage I with / without dW overlap in LinearFn"
for v in 0 1; do ACT_LINEAR_OVERLAP_DW=$v timeout 600 python bench.py --stage 1 --steps 20 --warmup 5 --no-cpu-baseline --no-instrument 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('LINEAR_OVERLAP_DW=$v', round(d['value'],1), round(d['ms_per_step'],3))"; done
for v in 0 1; do ACT_LINEAR_OVERLAP_DW=$v timeout 600 python bench.py --stage 3 --steps 40 --warmup 8 --no-cpu-baseline --no-instrument 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('finetune LINEAR_OVERLAP_DW=$v', round(d['value'],1), round(d['ms_per_step'],3))"; done
