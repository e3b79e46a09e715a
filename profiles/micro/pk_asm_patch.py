"""Edits the device assembly of feat_kernels.hip (SLP-vectorized build, -g1) for profiles/micro/pk_asm.sh: only instructions of
MfccKernel<512, 4> whose source line (.loc) lies in the power-spectrum loop are touched.  argv: <in.s> <out.s> <variant>
  none        unchanged
  nop_after   s_nop 7 behind every packed operation
  nop_before  s_nop 7 in front of every packed operation
  nop_mov     s_nop 7 in front of every 32-bit v_mov that follows a packed operation within three instructions
  unpack:<i>  packed operation number i of the region (in program order) replaced by its two 32-bit halves (only plain forms)"""
import re
import sys
src, dst, variant = sys.argv[1], sys.argv[2], sys.argv[3]
lines = open(src).read().split("\n")
# source lines of the loop (feat_kernels.hip): from "for (int k = lane + 1; 2 * k <= NC" to the "if (lane == 0)" behind it
hip = open(sys.argv[4]).read().split("\n")
lo = next(i for i, l in enumerate(hip) if "for (int k = lane + 1; 2 * k <= NC" in l) + 1
hi = next(i for i, l in enumerate(hip) if i > lo and "const float d0 = xr[m.fft_perm[0]]" in l)
def unpack(ins):
    """v_pk_{add,mul}_f32 vD, vA, (vB | constant) [op_sel:[a,b]] [op_sel_hi:[a,b]] [neg_lo:[0,x]] [neg_hi:[0,x]] as two 32-bit operations
    on the same registers, ordered so that neither reads a half the other has already written."""
    m = re.match(r"(v_pk_add_f32|v_pk_mul_f32)\s+v\[(\d+):\d+\],\s*v\[(\d+):\d+\],\s*(v\[(\d+):\d+\]|[-0-9.]+)(.*)$", ins)
    if not m:
        raise SystemExit("unpack: form not handled: " + ins)
    op, d, a, b_txt, b, mods = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4), m.group(5), m.group(6)
    def mod(name, default):
        mm = re.search(name + r":\[(\d),(\d)\]", mods)
        return [int(mm.group(1)), int(mm.group(2))] if mm else default
    op_sel, op_sel_hi, neg_lo, neg_hi = mod("op_sel", [0, 0]), mod("op_sel_hi", [1, 1]), mod("neg_lo", [0, 0]), mod("neg_hi", [0, 0])
    assert neg_lo[0] == 0 and neg_hi[0] == 0, ins
    def src1(sel):
        return b_txt if b is None else f"v{int(b) + sel}"
    def one(dst, s0_sel, s1_sel, neg):
        if op == "v_pk_mul_f32":
            assert not neg, ins
            return (f"v_mul_f32_e32 v{dst}, {src1(s1_sel)}, v{a + s0_sel}" if b is None else f"v_mul_f32_e32 v{dst}, v{a + s0_sel}, {src1(s1_sel)}"), {a + s0_sel} | ({int(b) + s1_sel} if b is not None else set())
        mn = "v_sub_f32_e32" if neg else "v_add_f32_e32"
        assert b is not None, ins
        return f"{mn} v{dst}, v{a + s0_sel}, {src1(s1_sel)}", {a + s0_sel, int(b) + s1_sel}
    lo_txt, lo_reads = one(d, op_sel[0], op_sel[1], neg_lo[1])
    hi_txt, hi_reads = one(d + 1, op_sel_hi[0], op_sel_hi[1], neg_hi[1])
    if d not in hi_reads:
        return [lo_txt, hi_txt]
    if d + 1 not in lo_reads:
        return [hi_txt, lo_txt]
    raise SystemExit("unpack: needs a temporary: " + ins)


out = []
in_kernel = False
cur_loc = 0
since_pk = 99
n_pk = 0
for l in lines:
    if l.startswith("_ZN2rs10MfccKernelILi512ELi4EE") and l.rstrip().endswith(":") or ("_ZN2rs10MfccKernelILi512ELi4EE" in l and l.lstrip().startswith("_ZN2rs10MfccKernelILi512ELi4EE")):
        in_kernel = True
    if in_kernel and "s_endpgm" in l:
        in_kernel = False
    m = re.match(r"\s*\.loc\s+\d+\s+(\d+)\s", l)
    if m and int(m.group(1)) != 0:          # (line 0 = compiler-generated: stays with the region it sits in)
        cur_loc = int(m.group(1))
    ins = l.strip()
    is_ins = in_kernel and ins and not ins.startswith((".", ";")) and not ins.endswith(":")
    in_region = in_kernel and lo <= cur_loc <= hi
    if is_ins and in_region and ins.startswith("v_pk_"):
        n_pk += 1
        if variant == "nop_before":
            out.append("\ts_nop 7")
        if variant.startswith("op1_") and n_pk == 1:
            # the one instruction the bisect singled out, in other forms (same products):
            #   v_pk_mul_f32 v[12:13], v[12:13], v[14:15] op_sel:[0,1] op_sel_hi:[0,0]      lo = v12 * v15, hi = v12 * v14
            assert ins.replace(" ", "") == "v_pk_mul_f32v[12:13],v[12:13],v[14:15]op_sel:[0,1]op_sel_hi:[0,0]", ins
            if variant == "op1_commuted":        # the pair that is overwritten as source 1
                out.append("\tv_pk_mul_f32 v[12:13], v[14:15], v[12:13] op_sel:[1,0] op_sel_hi:[0,0]")
            elif variant == "op1_copy_hi":       # low word copied into the high word first: no half reads across
                out.append("\tv_mov_b32_e32 v13, v12")
                out.append("\ts_nop 1")
                out.append("\tv_pk_mul_f32 v[12:13], v[12:13], v[14:15] op_sel:[0,1] op_sel_hi:[1,0]")
            elif variant == "op1_swapped_kn":    # the loaded pair swapped by a packed move first, then a plain in-place packed multiply with the low word splatted
                out.append("\tv_mov_b32_e32 v13, v12")
                out.append("\tv_pk_mov_b32 v[14:15], v[14:15], v[14:15] op_sel:[1,0]")
                out.append("\ts_nop 1")
                out.append("\tv_pk_mul_f32 v[12:13], v[12:13], v[14:15]")
                out.append("\ts_nop 1")
                out.append("\tv_pk_mov_b32 v[14:15], v[14:15], v[14:15] op_sel:[1,0]")
            elif variant == "op1_sleep":         # as built, a hundred-odd clocks behind the wait for the loads
                out.append("\ts_sleep 2")
                out.append(l)
            elif variant == "op1_selfmove":      # as built, the loaded pair passed through the VALU first (v_mov onto itself)
                out.append("\tv_mov_b32_e32 v14, v14")
                out.append("\tv_mov_b32_e32 v15, v15")
                out.append("\ts_nop 1")
                out.append(l)
            elif variant == "op1_raw_fence":     # as built, then an instruction that READS its result before anything overwrites source 1
                out.append(l)
                out.append("\ts_nop 0")
                out.append("\tv_mov_b32_e32 v13, v13")
                out.append("\tv_mov_b32_e32 v12, v12")
            elif variant == "op1_two_waits":     # as built, behind a second wait for the loads and eight wait states
                out.append("\ts_waitcnt vmcnt(0) lgkmcnt(0)")
                out.append("\ts_nop 7")
                out.append(l)
            else:
                raise SystemExit("unknown variant " + variant)
            since_pk = 0
            continue
        if variant.startswith("unpack:") and (variant == "unpack:all" or str(n_pk) in variant.split(":")[1].split(",")):
            for u in unpack(ins):
                out.append("\t" + u)
            since_pk = 0
            continue
        out.append(l)
        if variant == "nop_after":
            out.append("\ts_nop 7")
        since_pk = 0
        continue
    if is_ins:
        if variant == "nop_mov" and in_region and since_pk < 3 and ins.startswith("v_mov_b32"):
            out.append("\ts_nop 7")
        since_pk += 1
    out.append(l)
open(dst, "w").write("\n".join(out))
sys.stderr.write(f"{variant}: {n_pk} packed operations in source lines {lo}-{hi}\n")
