#!/bin/bash
O=gpurun_out/r04p; mkdir -p $O
run() { name=$1; shift; timeout 600 python3 bench.py --no-cpu-baseline "$@" 2> /dev/null | tail -1 > $O/bench_$name.json; python3 -c "
import json
d=json.load(open('$O/bench_$name.json')); print('$name', d['value'], d['roofline']['traffic'], d['roofline'].get('traffic_source'))"; }
run weightsfp8 --weights fp8
run fp16 --dtype fp16
run frames16weightsfp8 --frames 16 --weights fp8
