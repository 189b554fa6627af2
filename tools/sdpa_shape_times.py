"""Per-shape averages of the SDPA kernels from the kernel trace of tools/bench_sdpa_branch.py (three shapes x 11 calls, the first of each dropped):
    bash tools/prof_sdpa_branch.sh <tag>; python tools/sdpa_shape_times.py gpurun_out/prof_sdpab_<tag>/p_kernel_trace.csv"""
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
per=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    for k in ('sdpa_fwd','sdpa_bwd_delta','sdpa_bwd_dkv3','sdpa_bwd_dq2'):
        if k in n: per[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in per.items(): print("%-16s" % k, [round(sum(v[j*11+1:(j+1)*11])/10,1) for j in range(3)], "(16 plain x 2048 | 8 packed 3428 | 8 plain x 3428)")
