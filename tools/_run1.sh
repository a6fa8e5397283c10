timeout 600 python tools/ab_step.py ops.PIN_NEGATIVES False True 2>&1 | tail -3
