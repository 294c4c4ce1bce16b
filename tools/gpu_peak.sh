#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/conv_bench.py NONE > gpurun_out/peak.log 2>&1; cat gpurun_out/peak.log
