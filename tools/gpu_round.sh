#!/bin/bash
# full validation + kernel stats + HBM PMC passes in one box
bash tools/gpu_full.sh
bash tools/gpu_pmc_hbm.sh
