run() { echo "== $1 | $2 $3 | $4"; env $4 ACC_OVER="$1" python tools/acc_distribution.py $2 $3 2>&1 | tail -4 | cut -c1-105; }
run '{}' 2048 16 ACC_VAMP=3.0
run '{}' 512 32 ACC_VAMP=3.0
run '{}' 2048 16 "ACC_FLAT=0 ACC_Z=0.08"
run '{}' 512 32 "ACC_FLAT=0 ACC_Z=0.15"
run '{"obstacle":1,"obstacle_pos":[0.35,0.0,0.1]}' 1024 16 ACC_QAMP=1.2
run '{"hull_sides":0,"contact_model":0,"relative_breaking_threshold":0}' 1024 16 ACC_QAMP=1.0
