#!/bin/bash
# round 3, final run: full GPU suite + smoke, every bench line + kernel traces, PMC passes -- all at the final sources
bash tools/gpu_full.sh
bash tools/gpu_r3_evidence_b.sh
cp gpurun_out/hbm_traffic.json profiles/r03_hbm_traffic.json      # (on the box's copy: the bench lines below then carry traffic measured at these sources)
bash tools/gpu_r3_evidence_a.sh
