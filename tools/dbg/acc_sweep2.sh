run() { echo "== $1 | $2 $3"; ACC_OVER="$1" python tools/acc_distribution.py $2 $3 2>&1 | tail -4; }
run '{"cone_friction":0}' 2048 16
run '{"cone_friction":0}' 512 32
run '{"dt":0.01,"gravity_z":-9.81,"max_motor_impulse":0.04}' 2048 16
run '{"dt":0.01,"gravity_z":-9.81,"max_motor_impulse":0.04}' 512 32
run '{"n_iterations":10}' 2048 16
run '{"mu_link":0.5}' 2048 16
