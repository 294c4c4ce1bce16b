#!/bin/bash
# round 3, final evidence (b): PMC passes of the cfg3 bench command -- HBM traffic per kernel (FETCH_SIZE / WRITE_SIZE in separate passes)
# and MFMA-pipe utilisation
bash tools/gpu_pmc_hbm.sh
bash tools/gpu_pmc_sq.sh
