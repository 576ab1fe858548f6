#!/bin/bash
# usage: sweep_time.sh out per_gpu   -- the README recipe as a sweep on one GPU (20 canonical frames x 15 000 iterations, energies)
out=$1; pg=$2
rm -rf /tmp/sweep_recipe
timeout 900 python3 -m reart_amd.sweep --synthetic 1 --synthetic_frames 20 --cano all --n_iter 15000 --use_flow_loss --use_assign_loss --energy --per_gpu $pg --save_root /tmp/sweep_recipe > $out.line.json 2> $out.err
tail -2 $out.err
python3 -c "
import json; d=json.load(open('/tmp/sweep_recipe/sweep.json')); print('per_gpu $pg wall_s', d['wall_s'], 'it/s', d['iterations_per_s'], d['rank0_stages'])"
cp /tmp/sweep_recipe/sweep.json $out.sweep.json
