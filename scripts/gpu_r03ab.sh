#!/bin/bash
# round 3, last full GPU pass: gpu tests, bench line, traces, counters (batch 256) + the float16x3 forward's counters at batch 512 / 1024
bash scripts/gpu_round.sh r03ab tests pmc
bash scripts/pmc_batches.sh r03ab float16x3
