run() { echo "== $1 | $2 $3"; ACC_OVER="$1" python tools/acc_distribution.py $2 $3 2>&1 | tail -4; }
run '{"hull_sides":0,"contact_model":0,"relative_breaking_threshold":0}' 2048 16
run '{"hull_sides":0,"contact_model":0,"relative_breaking_threshold":0}' 512 32
run '{"warm_start":1}' 2048 16
run '{"obstacle":1,"obstacle_pos":[0.35,0.0,0.1]}' 2048 16
run '{"obstacle":2,"obstacle_pos":[0.35,0.0,0.1]}' 1024 16
run '{"obstacle":1,"obstacle_pos":[0.35,0.0,0.1]}' 512 32
