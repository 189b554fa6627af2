"""Build-time check of sdpa_fwd3's generated-asm kernel (device assembly from `hipcc -S --cuda-device-only sdpa.hip`):
  * no compiler-generated instruction touches a128-a191: the Q fragments are fetched by hand into those registers (behind a block, for the NEXT
    row block) and may still be in flight while the compiler's code runs - a copy would read them too early;
  * no compiler-generated wait for vector memory (s_waitcnt vmcnt) and no vector load outside the asm statements on the common path: the compiler
    cannot count past an asm block, so any wait of its own is a wait for everything the blocks have in flight (reported, not fatal: the rare
    repeat path may hold some).
Run by halva_amd/csrc/Makefile on the device assembly of the same command line that builds sdpa.o; a touched Q register fails the build.
usage: python tools/check_fwd3_isa.py <file.s>"""
import re, sys
txt = open(sys.argv[1]).read().split("\n")
starts = [i for i, l in enumerate(txt) if re.match(r"_ZN\S*sdpa_fwd3_kernel\S*:", l)]
assert starts, "no sdpa_fwd3 kernel in " + sys.argv[1]
bad = notes = 0
for start in starts:
    end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
    inasm = False
    for i in range(start, end):
        l = txt[i]
        if "ASMSTART" in l: inasm = True; continue
        if "ASMEND" in l: inasm = False; continue
        if inasm or l.strip().startswith(";"): continue
        code = l.split(";")[0]
        for m in re.finditer(r"\ba\[?(\d+)(?::(\d+))?\]?", code):
            lo = int(m.group(1)); hi = int(m.group(2) or lo)
            if hi >= 128 and lo <= 191:
                print("Q fragment register touched by the compiler:", l.strip()); bad += 1
        if "scratch_" in code or "s_waitcnt vmcnt" in code or re.search(r"\b(global|buffer|flat)_load", code):
            notes += 1
            if "-v" in sys.argv: print("note:", l.strip())
print("sdpa_fwd3 ISA check:", "FAILED (%d)" % bad if bad else "ok", "(%d compiler-side scratch / vector-load / vmcnt lines: see -v)" % notes)
sys.exit(1 if bad else 0)
