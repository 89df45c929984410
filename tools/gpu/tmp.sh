#!/bin/bash
cd /root/repo
for i in 1 2; do
for e in 512 256; do
echo "DY_WG3_BLOCKS=$e"
DY_WG3_BLOCKS=$e timeout -k 10 600 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-160
done
done
