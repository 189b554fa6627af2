"""Build-time check of sdpa_bwd_dkv3's generated-asm kernel (device assembly from `hipcc -S --cuda-device-only sdpa.hip`):
  * no compiler-generated instruction touches a128-a191 (the K / V fragments are fetched by hand into those registers and may still be in
    flight while the compiler's code runs: a copy would read them too early);
  * no scratch (a scratch reload waits, in order, for every tile request in flight);
  * no FLAT instruction (round 6: the dK / dV row pointers come out of the item record as integers; stored through a generic pointer they were
    flat_store_dwordx4, which count on lgkmcnt too - every LDS read behind one then waited for vmcnt(0)).
  (round 6: the fragments are literal registers in the asm statements, not operands: the rule is the ONLY thing that keeps them safe, together
  with the clobber list of the block, which keeps the compiler's spills out of them.)
(a0-a127, the accumulators, are NOT checked: the compiler legitimately reads them for the store tail behind the last asm statement.)
  (The RoPE tables of the dK store epilogue travel through LDS since round 5 - dkv3_rope_request_lds, sdpa_dkv3.h - and touch no register.)
Run by halva_amd/csrc/Makefile on the device assembly of the same command line that builds sdpa.o; a failure fails the build.
usage: python tools/check_dkv3_isa.py <file.s>"""
import re, sys
txt = open(sys.argv[1]).read().split("\n")
bad = 0
starts = [i for i, l in enumerate(txt) if re.match(r"_ZN\S*sdpa_bwd_dkv3_kernelILi128ELb[01]ELb1\S*:", l)]      # the ASM = true instantiations
assert starts, "no sdpa_bwd_dkv3 kernel in " + sys.argv[1]
for start in starts:
    end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
    inasm = False
    for i in range(start, end):
        l = txt[i]
        if "ASMSTART" in l: inasm = True; continue
        if "ASMEND" in l: inasm = False; continue
        if inasm or l.strip().startswith(";"): continue
        if "scratch_" in l:
            print("scratch:", l.strip()); bad += 1
        if re.match(r"\s*flat_", l):      # (round 6) a generic-pointer access: FLAT instructions count on the LDS counter as well - measured 7 000 cycles per item
            print("FLAT instruction (address built from integers without a global-typed pointer?):", l.strip()); bad += 1
        for m in re.finditer(r"\ba\[?(\d+)(?::(\d+))?\]?", l.split(";")[0]):
            lo = int(m.group(1)); hi = int(m.group(2) or lo)
            if hi >= 128 and lo <= 191:
                print("K/V fragment register touched by the compiler:", l.strip()); bad += 1
print("sdpa_bwd_dkv3 ISA check:", "FAILED (%d)" % bad if bad else "ok")
sys.exit(1 if bad else 0)
