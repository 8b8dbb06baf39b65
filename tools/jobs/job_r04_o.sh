# round 4, job o: planes TN GEMM inside the step: serial (no side stream), fewer row splits
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_o
mkdir -p $O
b() { python bench.py --config $1 --step-only --steps $2 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.load(sys.stdin); print(round(d["ms_per_step"],3), round(d["ms_per_step_median"],3))'; }
for v in "X=1" "GCPNET_TN_PLANES=1" "GCPNET_SIDE_STREAM=0" "GCPNET_SIDE_STREAM=0 GCPNET_TN_PLANES=1" "GCPNET_TN_PLANES=1 GCPNET_TN_SPLITS=64" "GCPNET_TN_PLANES=1 GCPNET_TN_SPLITS=32" "GCPNET_TN_PLANES=1 GCPNET_TN_SPLITS=256" "GCPNET_SIDE_STREAM=0 GCPNET_TN_PLANES=1 GCPNET_TN_SPLITS=256"; do
  echo "c2 [$v] $(env $v bash -c "$(declare -f b); b c2 20")" >> $O/step.txt
done
for v in "X=1" "GCPNET_SIDE_STREAM=0 GCPNET_TN_PLANES=1" "GCPNET_SIDE_STREAM=0 GCPNET_TN_PLANES=1 GCPNET_TN_SPLITS=256"; do
  echo "c5 [$v] $(env $v bash -c "$(declare -f b); b c5 4")" >> $O/step.txt
done
cat $O/step.txt
