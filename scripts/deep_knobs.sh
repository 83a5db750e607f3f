#!/bin/bash
# deep rows of bench.py under a few settings (round 4: de-phased workgroups, 13-qubit tiles at n = 20)
bash scripts/deep_ab.sh "QSV_DEPHASE=0" "QSV_DEPHASE=1" "QSV_DEPHASE=2" "QSV_DEPHASE=3" "QSV_DEPHASE=5"
ROWS=deep_n20_L8 bash scripts/deep_ab.sh "QSV_TILE_BITS=13" "QSV_TILE_BITS=13 QSV_REG_BITS=4" "QSV_TILE_BITS=12 QSV_REG_BITS=3"
