#!/bin/bash
# round 5: spread of the training pins under MIOpen's two find modes (FAST = heuristic choice, DYNAMIC_HYBRID = timed choice)
mkdir -p gpurun_out/pins_spread
for mode in FAST DYNAMIC_HYBRID; do
  for rep in 1 2; do
    rm -f gpurun_out/training_pins_measured.json
    SP3D_PINS_FIND_MODE=$mode python -m pytest tests/test_gpu_reference_pins_r3.py -m gpu -q -k "ssv_train_step or (supervised and batched and False)" 2>&1 | tail -3
    cp gpurun_out/training_pins_measured.json gpurun_out/pins_spread/${mode}_${rep}.json
  done
done
