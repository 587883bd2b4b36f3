for bf in "" "--bf16"; do CB_JOIN_PROBE=1 python tools/ab_step.py --only $bf --steps 6 --rounds 1 --tag probe 2>&1 | grep "join probe" | tail -8; done
