#!/bin/bash
export VARIANTS="cur nostat nodma cur nostat" SKIPTESTS=1
bash tools/r04/g20.sh
