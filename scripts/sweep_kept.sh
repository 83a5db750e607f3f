for cfg in "QSV_GROUP=128" "QSV_GROUP=32" "QSV_GROUP=16" "QSV_GROUP=8" "QSV_GROUP=16 QSV_PUSH_EVALS=8" "QSV_GROUP=32 QSV_PUSH_EVALS=16" "QSV_GROUP=8 QSV_PUSH_EVALS=4" "QSV_GROUP=24 QSV_PUSH_EVALS=12" "QSV_GROUP=128 QSV_STREAMS=1" "QSV_GROUP=16 QSV_STREAMS=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python scripts/prefix_cache_experiment.py 20:8:64 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        if d['searched_layer'] == 7: print('  full %d kept %d mixed %d' % (d['full_evals_per_s'], d['kept_evals_per_s'], d['mixed_evals_per_s']))
"
done
