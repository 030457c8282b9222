#!/bin/bash
cd /root/repo
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 | cut -c1-110; done
