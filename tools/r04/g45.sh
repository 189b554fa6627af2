#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/pmc_sdpa.sh r04 > /dev/null 2>&1; cp gpurun_out/pmc_sdpa_r04/summary.json gpurun_out/r04_sdpa_all_pmc.json; rm -rf gpurun_out/pmc_sdpa_r04
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04_sdpa_all_pmc.json'))
for k,v in d.items(): print(k, {a:(round(b,4) if b<10 else int(b)) for a,b in v['derived'].items()})
PY
