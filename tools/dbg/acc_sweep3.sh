run() { echo "== $1 | $2 $3 | $4"; env $4 ACC_OVER="$1" python tools/acc_distribution.py $2 $3 2>&1 | tail -4 | cut -c1-105; }
run '{"warm_start":1}' 512 32 X=1
run '{"warm_start":1,"obstacle":2,"obstacle_pos":[0.35,0.0,0.1]}' 1024 16 X=1
run '{"hull_sides":8}' 2048 16 X=1
run '{"hull_sides":8}' 512 32 X=1
run '{"inertia_from_file":1}' 2048 16 X=1
run '{}' 2048 16 ACC_MU=1
run '{}' 512 32 ACC_MU=1
run '{}' 2048 16 ACC_QAMP=1.7
run '{}' 512 32 ACC_QAMP=1.7
