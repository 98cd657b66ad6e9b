#!/bin/bash
# r06zc: SQ / GRBM / TCC counter passes over the final library's 256-clip step (MFMA busy, LDS, waits, L2 hit rates per kernel), and over the 3-clip step (the skinny kernel)
bash tools/pmc_sq.sh r06zc
bash tools/pmc_sq.sh r06zc_B3 --batch 3
