# Force row programs of s16h74l4c4 with DNAS_PLAN_PICK=rows,S-rows,groups,ascending,plain,typedS and time them
# (bench.py --reads 2880 --timed-only): does the plan's cost model pick the fastest?   bash tools/plan_pick_scan.sh [picks...]
PICKS=${@:-"14,5,1,1,0,1 14,5,1,0,1,1 14,5,3,0,1,1 14,5,2,0,1,1 14,5,1,1,0,0 14,5,1,0,1,0 13,5,1,0,1,1 14,5,3,0,1,0 14,5,2,0,1,0"}
for pick in $PICKS; do
  DNAS_AUTOTUNE=0 DNAS_PLAN_PICK=$pick timeout -k 10 200 python bench.py --reads 2880 --cpu-seconds 0 --timed-only --steps 2 > gpurun_out/pick.log 2>&1 || { echo "pick $pick failed"; tail -2 gpurun_out/pick.log; continue; }
  tail -1 gpurun_out/pick.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pick $pick', round(d['value']), round(d['roofline']['frac'],4), round(d['roofline'].get('rounds_per_column',0),2))"
done
