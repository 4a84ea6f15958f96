#!/bin/bash
# one short bench line on this box (for the range over boxes quoted in README): the driver's default minus the slow extras
python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-configs --alt-batch 0 2>/dev/null | grep "^{" | tail -1
