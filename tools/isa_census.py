#!/usr/bin/env python3
"""Cycle-weighted opcode census of the path megakernel (no GPU needed).

What it does
  1. compiles csrc/ssx_kernels.hip for gfx950 with line tables (-g) into a device object;
  2. disassembles one kernel (llvm-objdump) and asks llvm-symbolizer for the INLINE CHAIN of every instruction
     address, so that a helper inlined in several places (shear_xyz, ssx_exact::rcp, dot3 ...) is attributed to the
     call site it was inlined into -- e.g. [div64_rcp_any < ray_setup < trace < shadow_flush < flush_fold < render_body];
  3. maps every chain to a REGION of the kernel (pass 1 / pass 2 / ray_setup of the primary and of the shadow trace,
     sphtri_make, Arvo's sampler, albedo, cosine sampler, level logging, fold levels, fold pass, unit hand-over ...);
  4. weights each static instruction with its region's wave-level executions per loop iteration of the path kernel,
     taken from the lane-occupancy build's counters (tools/lanestat.py -> profiles/*/lanestat.log: "entries" are
     wave-level executions of the region's SSX_STAT site);
  5. prices every VALU opcode with its measured issue cost at four waves per SIMD (tools/ubench/valu_rates ->
     profiles/*valu_rates.log), operand forms included (an SGPR source, a third VGPR source, a constant operand);
  6. prints region x class tables of static count, dynamic count per iteration, cycles per iteration, share of the
     VALU time and mean lane occupancy, and the top opcodes per region.

Static counts x measured trip counts are a model, not a trace: branches inside a region that the counters do not see
(rare fallbacks: sphtri_make_general, the ambiguous-sine fallback, Lemire's rejection loop, other uplifts) are priced
at the weight RARE below and listed as "cold".  The model's total is printed next to the measured cycles per iteration
(SQ_BUSY_CYCLES-free estimate: kernel time x clock x SIMDs / iterations) so that its error is visible.

    python tools/isa_census.py [--kernel ssx_render_kernel_cornell] [--lanestat profiles/r04/lanestat.log]
                               [--rates profiles/r02_valu_rates.log] [--csv profiles/r05/isa_census.csv] [-D MACRO ...]
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.environ.get("SSX_CENSUS_SRC") or os.path.join(ROOT, "simple_spectral_amd", "csrc", "ssx_kernels.hip")   # (another revision's source: a worktree)
LLVM = "/opt/rocm/lib/llvm/bin"
RARE = 0.002  # weight of code behind a branch that the counters do not see and the design calls rare
UNIT_PASSES = 2.0  # fold passes per work unit: 2 for units of four samples per pixel (Cornell), 4 for units of eight (plane-srgb): --unit-passes


# ------------------------------------------------------------------------------------------------ disassembly
def build_object(defs, out):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-g", "-c",
           "--cuda-device-only", "--no-gpu-bundle-output", "-o", out, SRC] + ["-D" + d for d in defs]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)


def disassemble(obj, kernel):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + kernel, obj], text=True)
    ins = []
    for ln in txt.split("\n"):
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return ins


def inline_chains(obj, addrs):
    """[(function, line), ...] innermost first, for every address."""
    p = subprocess.run([os.path.join(LLVM, "llvm-symbolizer"), "--obj=" + obj, "--inlines", "--functions=short"],
                       input="\n".join("0x%x" % a for a in addrs) + "\n", capture_output=True, text=True, check=True)
    chains, cur, pending = [], [], None
    for ln in p.stdout.split("\n"):
        if ln == "":
            if cur or pending:
                chains.append(cur)
            cur, pending = [], None
            continue
        if pending is None:
            pending = ln
        else:
            m = re.match(r"(.*):(\d+):(\d+)$", ln)
            cur.append((pending, os.path.basename(m.group(1)) if m else "?", int(m.group(2)) if m else 0))
            pending = None
    assert len(chains) == len(addrs), (len(chains), len(addrs))
    return chains


# ------------------------------------------------------------------------------------------------ regions
def line_of(chain, func):
    """line inside `func`'s own body that this instruction belongs to (the call line if it comes from a callee)"""
    for f, _, ln in chain:
        if f == func or f.startswith(func + "<"):
            return ln
    return None


def has(chain, *funcs):
    names = [f for f, _, _ in chain]
    return any(n == g or n.startswith(g + "<") for n in names for g in funcs)


def source_ranges():
    """line ranges of the lambdas and loops of render_body / resolve_records / trace, found by their first lines' text"""
    lines = open(SRC).read().split("\n")

    def find(pat, start=0):
        for i in range(start, len(lines)):
            if pat in lines[i]:
                return i + 1
        raise KeyError(pat)

    def block_end(first):  # matching brace of the block opened on line `first`
        depth, started = 0, False
        for i in range(first - 1, len(lines)):
            code = re.sub(r"//.*", "", lines[i])
            depth += code.count("{") - code.count("}")
            started = started or "{" in code
            if started and depth <= 0:
                return i + 1
        return len(lines)

    r = {}
    for name, pat in (("end_path", "auto end_path = [&]"), ("refill", "auto refill = [&]"), ("flush_fold", "auto flush_fold = [&]"),
                      ("rotate_fetch", "auto rotate_fetch = [&]")):
        a = find(pat)
        r[name] = (a, block_end(a))
    a = find("if (!cur_valid && more) {")
    r["unit_fetch"] = (a, block_end(a))
    a = find("while (cand) {")
    r["pass2"] = (a, block_end(a))
    a = find("for (uint32_t d = top; d-- > 0u;) {")
    r["fold_levels"] = (a, block_end(a))
    a = find("if ((els ? (p.depth == 0u")
    r["emission"] = (a, block_end(a))
    a = find("if (els && (!a.indirect_only || p.depth > 0u)) {")
    r["nee"] = (a, block_end(a))
    a = find("if (n_dot_l > 0.0f) {", r["nee"][0])
    r["nee_contrib"] = (a, block_end(a))
    a = find("// the factors of the continuation for the backward fold")
    r["log_level"] = (a, find("return true;", a))
    a = find("// (4)")
    r["main_trace"] = (a, find("if (a.pre_hits) refill(true);"))
    # single lines / blocks behind branches that the built-in Cornell configuration (almost) never takes: (function, first, last)
    cold = []
    def one(func, pat, block=False, start=0):
        a = find(pat, start)
        cold.append((func, a, block_end(a) if block else a))
    one("normalize3_any", "if (d < 0x1p-100f) s = 1.0f / __builtin_sqrtf(d);")
    one("func_bar", "if (lensq < 0x1p-100f) is = 1.0f / __builtin_sqrtf(lensq);")
    try:
        one("trace", "__builtin_fabsf(W)) == 0.0f) {", True)
    except KeyError:
        one("trace", "if (U == 0.0f || V == 0.0f || W == 0.0f) {", True)     # (round 4's source)                      # the binary64 edge fallback (its three compares are priced below)
    one("operator()", "else { const uint32_t r = item % npx;")                          # ragged tiles
    one("render_body", "else { const uint32_t r = item % npx;")
    one("resolve_records", "if ((K[s] >> 26) & 1u) {", True)
    one("resolve_records", "if (no_flat_field) {", True)
    one("resolve_records", "if (rgb_mode) { xyz[0] = rad[s][0];")
    one("resolve_records", "if (keep_samples) a.ray[r0 + s * stride]")
    one("flux_to_xyz", "bar[0] = spectrum_hero(L, h.spec_xbar")
    one("flux_to_xyz", "bar[1] = spectrum_hero(L, h.spec_ybar")
    one("flux_to_xyz", "bar[2] = spectrum_hero(L, h.spec_zbar")
    one("material_albedo", "if (Q.albedo_mode == 0u) return spectrum_hero(L, Q.albedo")
    one("material_albedo", "return texture_sample(L, Q.albedo_tex")
    a = find("w_i = reflect3(mk(-p.dir.x")
    cold.append(("path_step", a, a + 4))                                                  # MaterialMirror
    one("sample_light", "if (nl == 1u) (void)rng_next(rng); else pick = rand_choice(rng, nl);")  # priced at its one-light cost below (see RNG_LINE)
    r["_cold"] = cold
    r["_stage"] = find("if (stage) stage_sample(")
    return r


def region_of(chain, R):
    """-> (region name, counter key) ; counter keys are resolved to per-iteration weights by weights()"""
    def within(func, key):
        ln = line_of(chain, func)
        return ln is not None and R[key][0] <= ln <= R[key][1]

    cold = has(chain, "sphtri_make_general", "jh_uplift", "meng_uplift", "texture_sample", "ssx_cosf_lds", "rand_choice", "stage_sample", "add_staged_sample")
    for func, a, b in R["_cold"]:
        ln = line_of(chain, func)
        if ln is not None and a <= ln <= b and not (func == "trace" and ln == a) and not (func == "sample_light" and not has(chain, "rand_choice")):
            cold = True
    if cold and not has(chain, "path_step"):
        return "cold (rare fallbacks / other uplifts / modes)", "rare"
    if has(chain, "sphtri_make") and has(chain, "ssx_sinf_lds"):
        cold = True  # the ambiguous-sine fallback (~2^-11 per call)
    if has(chain, "unit_fold", "resolve_records", "sums_chain"):
        if has(chain, "sums_chain"):
            return "fold: parked units chain", "parked"
        if has(chain, "resolve_records"):
            if within("resolve_records", "fold_levels"):
                return "fold: level step (2 ways)", "fold_level"
            if has(chain, "flux_to_xyz", "add_sample", "stage_sample"):
                return "fold: flux->XYZ + pixel sums", "fold_pass"
            return "fold: pass set-up (tails, last level)", "fold_pass"
        return "fold: unit hand-over", "unit"
    if has(chain, "trace"):
        shadow = has(chain, "shadow_flush")
        who = "shadow" if shadow else "primary"
        if has(chain, "ray_setup"):
            return "trace %s: ray_setup" % who, "trace_" + who
        if has(chain, "pass1_cornell", "pass1_plane", "pass1_jit", "quad_flags"):
            return "trace %s: pass 1" % who, "trace_" + who
        if within("trace", "pass2"):
            return "trace %s: pass 2 trip" % who, "pass2_" + who
        return "trace %s: masks, candidates" % who, "trace_" + who
    if has(chain, "shadow_flush"):
        return "shadow flush: queue read, result store", "trace_shadow"
    if has(chain, "path_step"):
        if cold:
            return "cold (rare fallbacks / other uplifts / modes)", "rare"
        if has(chain, "sphtri_make"):
            if has(chain, "ssx_acos_sin_lds", "ssx_acosf_lds", "ssx_fm_asin_poly"):
                return "light: sphtri_make acos/sin (binary64)", "light"
            return "light: sphtri_make f32 part", "light"
        if has(chain, "rand_toward_sphericaltri"):
            if has(chain, "ssx_sincosf", "ssx_fm_reduce", "ssx_fm_ksin", "ssx_fm_kcos"):
                return "light: Arvo sincos (binary64)", "light"
            return "light: Arvo f32 part", "light"
        if has(chain, "skip_light_draws", "skip_coshemi_draws"):
            return "black surface: the samplers' draws only", "black"
        if has(chain, "sample_light"):
            return "light: pick, normalize x3, pdf", "light"
        if has(chain, "material_albedo"):
            if has(chain, "texel_lrgb", "hero_gather3"):
                return "albedo: textured branch", "albedo_tex"
            return "albedo: index + constant gather", "iter"
        if has(chain, "rand_coshemi"):
            if has(chain, "ssx_sincosf", "ssx_fm_reduce", "ssx_fm_ksin", "ssx_fm_kcos"):
                return "bsdf: coshemi sincos (binary64)", "bsdf"
            return "bsdf: coshemi f32 part", "bsdf"
        if has(chain, "get_rotated_to"):
            return "bsdf: get_rotated_to", "bsdf"
        if within("path_step", "emission"):
            return "emission lookup (camera hit on a light)", "emission"
        if within("path_step", "nee_contrib"):
            return "nee: contribution, park shadow ray", "nee"
        if within("path_step", "nee"):
            return "nee: n.l", "iter"
        if within("path_step", "log_level"):
            return "log level entry", "cont"
        return "path_step: rest (normal, f_lamb, continue test)", "iter"
    # render_body itself
    if has(chain, "generate_sample", "camera_dir"):
        return "loop: refill: sample made in the loop (fused generation)", "refill"
    for name, key in (("refill", "refill"), ("end_path", "iter"), ("flush_fold", "iter")):
        if within("render_body", name) or has(chain, "operator()") and within("operator()", name):
            return "loop: " + name, key
    if within("render_body", "unit_fetch") or (has(chain, "operator()") and within("operator()", "unit_fetch")) or has(chain, "unit_setup", "tile_of_slot"):
        return "loop: unit fetch + set-up", "unit"
    if within("render_body", "rotate_fetch") or (has(chain, "operator()") and within("operator()", "rotate_fetch")):
        return "loop: rotate", "iter"
    if has(chain, "hit_st"):
        return "loop: hit st (textured quad)", "iter"
    if has(chain, "stage_lds"):
        return "prologue (once per wave)", "once"
    ln = line_of(chain, "render_body")
    if ln is not None and R["main_trace"][0] <= ln <= R["main_trace"][1]:
        return "loop: after trace (hit -> path state)", "iter"
    return "loop: other", "iter"


def weights(lanestat_path):
    """wave-level executions per path-loop iteration, from the counters of the lane-occupancy build"""
    ent, lanes = {}, {}
    for ln in open(lanestat_path):
        m = re.match(r"^(.*?)\s{2,}(\d+)\s+([\d.]+)\s+([\d.]+)%", ln)
        if m:
            ent[m.group(1).strip()] = int(m.group(2))
            lanes[m.group(1).strip()] = float(m.group(3))
    it = ent["iteration: lanes with a path"]
    ways = 2.0  # SSX_RESOLVE_WAYS: the counter sits inside the unrolled way loop
    passes = ent["flux -> XYZ"] / ways
    e = lambda name, default=0.0: ent.get(name, default)   # (a scene without the region -- plane-srgb has no emission lookup, no camera pre-trace -- has no line)
    w = {
        "iter": 1.0, "once": 0.0, "rare": RARE,
        "trace_primary": e("primary trace: lanes with a ray") / it,
        "pass2_primary": e("primary pass-2 trips") / it,
        "trace_shadow": e("shadow trace: lanes with a ray") / it,
        "pass2_shadow": e("shadow pass-2 trips") / it,
        "albedo_tex": e("albedo: texture") / it,
        "emission": e("emission lookup") / it,
        "light": e("light sampling", it) / it,
        "bsdf": e("BSDF sample", it) / it,
        "black": e("black surface: draws only") / it,
        "refill": e("refill: lanes taking a sample", it) / it,     # wave-level executions of the refill's body (round 6's counter; before: once per iteration)
        "nee": e("NEE contribution") / it,
        "cont": e("continue (store fs/np)") / it,
        "fold_level": e("fold level x way") / ways / it,
        "fold_pass": passes / it,
        "unit": passes / UNIT_PASSES / it,      # a unit of four (eight) samples per pixel is two (four) passes (cohorts of two)
        "parked": 0.05 * passes / UNIT_PASSES / it,
    }
    l = lambda name: lanes.get(name, 64.0)
    occ = {
        "iter": l("iteration: lanes with a path"), "trace_primary": l("primary trace: lanes with a ray"),
        "pass2_primary": l("primary pass-2 trips"), "trace_shadow": l("shadow trace: lanes with a ray"),
        "pass2_shadow": l("shadow pass-2 trips"), "albedo_tex": l("albedo: texture"), "emission": l("emission lookup"),
        "light": l("light sampling"), "bsdf": l("BSDF sample"), "black": l("black surface: draws only"), "refill": l("refill: lanes taking a sample"),
        "nee": l("NEE contribution"), "cont": l("continue (store fs/np)"), "fold_level": l("fold level x way"),
        "fold_pass": 64.0, "unit": 64.0, "parked": 64.0, "rare": 1.0, "once": 64.0,
    }
    return w, occ, it


# ------------------------------------------------------------------------------------------------ prices
FAST = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_fmamk_f32", "v_fmaak_f32", "v_not_b32", "v_accvgpr")
SHIFT = ("v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32")


def load_rates(path):
    r = {}
    if path and os.path.exists(path):
        for ln in open(path):
            m = re.match(r"^\s+(.*?)\s+[\d.]+ ms\s+->\s+([\d.]+) cyc/instr", ln)
            if m:
                r[m.group(1).strip()] = float(m.group(2))
    return r


def price(op, operands, rates):
    """(cycles, class) of one VALU wave-instruction at 4 waves per SIMD"""
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    fast, slow = rates.get("v_add_f32", 2.6), rates.get("v_med3_f32", 4.3)
    ops = [o.strip() for o in operands.split(",")] if operands else []
    srcs = ops[1:]
    n_vgpr = sum(1 for o in srcs if re.match(r"^-?\|?v\[?\d", o))
    sgpr_src = any(re.match(r"^-?\|?s\[?\d", o) or o in ("vcc", "exec", "vcc_lo", "vcc_hi") for o in srcs) and not base.startswith(("v_cndmask", "v_readlane", "v_writelane", "v_div_fmas"))
    if base.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return rates.get("v_rcp_f64", 16.3), "trans f64"
    if base.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return rates.get("v_rcp_f32", 8.3), "trans f32"
    if "_f64" in base or base in ("v_lshl_add_u64", "v_lshlrev_b64", "v_lshrrev_b64", "v_mov_b64", "v_mad_u64_u32", "v_mad_i64_i32"):
        if base.startswith("v_cvt"):
            return rates.get("v_cvt_f64_f32", 4.2), "convert"
        if base in ("v_mov_b64",):
            return slow, "move"
        if "_f64" in base and not base.startswith(("v_cmp", "v_cndmask")):
            return rates.get("v_fma_f64", 4.3), "binary64 arith"
        if base.startswith("v_cmp"):
            return rates.get("v_cmp_lt_f32_e32 ->vcc", 4.24), "compare"
        return slow, "int 64-bit / addresses"
    if base.startswith("v_cmp"):
        return rates.get("v_cmp_lt_f32_e64 ->sgpr" if op.endswith("e64") else "v_cmp_lt_f32_e32 ->vcc", 4.3), "compare"
    if base.startswith("v_cndmask"):
        # VOP2 form behind its compare: 6.49 - 4.24 = 2.25; VOP3 form (SGPR-pair mask): 4.27
        return (rates.get("v_cndmask_b32_e64 sgpr", 4.27) if op.endswith("e64") else rates.get("v_cmp+v_cndmask (2 instr)", 6.49) - rates.get("v_cmp_lt_f32_e32 ->vcc", 4.24)), "select"
    if base.startswith(("v_min", "v_max", "v_med3")):
        return slow, "min/max/med3"
    if base.startswith("v_cvt") or base in ("v_floor_f32", "v_ceil_f32", "v_trunc_f32", "v_rndne_f32", "v_fract_f32"):
        return rates.get("v_cvt_i32_f32", 4.27), "convert"
    if base in ("v_fma_f32", "v_fmac_f32", "v_mad_f32"):
        if base == "v_fma_f32" and n_vgpr <= 2 and not sgpr_src:
            return rates.get("v_fma_f32 v,v,0", 2.53), "f32 fma"
        return rates.get("v_fma_f32", 3.96), "f32 fma"
    if base.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")):
        return rates.get("v_div_fixup_f32", 4.29), "division fix-ups"
    if base.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_mbcnt", "v_bcnt")):
        return slow, "lane ops"
    if base in FAST or base in SHIFT:
        if base in SHIFT and n_vgpr > 1:
            return rates.get("v_lshlrev_b32", 4.19), "int / bit ops (slow forms)"
        if sgpr_src:
            return rates.get("v_mul_f32 sgpr src", 4.25), ("f32 add/mul, SGPR source" if "_f32" in base else "int / bit ops (slow forms)" if base != "v_mov_b32" else "move")
        if base == "v_mov_b32":
            return rates.get("v_mov_b32", 2.41), "move"
        if "_f32" in base:
            return (rates.get("v_mul_f32", 2.64) if "mul" in base or "fma" in base else rates.get("v_add_f32", 2.8)), "f32 add/mul"
        return rates.get("v_and_b32", 2.5), "int / bit ops (fast forms)"
    return slow, "int / bit ops (slow forms)"


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="ssx_render_kernel_cornell")
    ap.add_argument("--lanestat", default=os.path.join(ROOT, "profiles", "r04", "lanestat.log"))
    ap.add_argument("--rates", default=os.path.join(ROOT, "profiles", "r02_valu_rates.log"))
    ap.add_argument("--csv", default="")
    ap.add_argument("--ops", type=int, default=6, help="top opcodes listed per region")
    ap.add_argument("--kernel-ms", type=float, default=20.1, help="measured duration of the launch the counters belong to, scaled to its samples")
    ap.add_argument("--samples", type=float, default=512 * 512 * 256)
    ap.add_argument("--lanestat-samples", type=float, default=512 * 512 * 64)
    ap.add_argument("--clock-mhz", type=float, default=2400.0)
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--obj", default="", help="reuse / keep the object here")
    ap.add_argument("--unit-passes", type=float, default=2.0, help="fold passes per work unit (2: units of four samples per pixel; 4: of eight, plane-srgb)")
    args = ap.parse_args()
    global UNIT_PASSES
    UNIT_PASSES = args.unit_passes

    R = source_ranges()
    with tempfile.TemporaryDirectory() as td:
        obj = args.obj or os.path.join(td, "k.o")
        if not (args.obj and os.path.exists(obj)):
            build_object(([] if ("plane" in args.kernel or args.kernel.endswith("_nq")) else ["SSX_PROBE_BUILD"]) + args.D, obj)   # (the probe build holds the generic and the Cornell kernel only)
        ins = disassemble(obj, args.kernel)
        chains = inline_chains(obj, [a for a, _, _ in ins])
    rates = load_rates(args.rates)
    W, OCC, iters_ls = weights(args.lanestat)

    # The pair-aware model (profiles/r05/valu_rates.log, "pair" rows): an instruction of the 4.3-cycle kind next to one of the 2.5-cycle
    # kind costs the pair ~4.7 cycles, two of the slow kind 8.5, a binary64 operation does not overlap with anything (mul | fma_f64 6.9):
    # per region  max(2.35 x (fast + slow), 4.2 x slow) + 4.3 x binary64 + 8.8 x transcendental f32 + 16.3 x transcendental f64.
    pair = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0.0, 0.0])   # region -> dynamic [fast, slow, f64, trans32, trans64]
    table = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))   # region -> class -> [static, dyn, cycles]
    opsum = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))   # region -> opcode form -> ...
    key_of = {}
    n_valu = n_other = 0
    for (addr, op, operands), chain in zip(ins, chains):
        if not op.startswith("v_"):
            n_other += 1
            continue
        n_valu += 1
        region, key = region_of(chain, R)
        key_of[region] = key
        cyc, cls = price(op, operands, rates)
        w = W[key]
        pair[region][4 if cls == "trans f64" else 3 if cls == "trans f32" else 2 if cls in ("binary64 arith",) or (cls == "convert" and "f64" in op) else 1 if cyc > 3.5 else 0] += w
        for t, k in ((table[region], cls), (opsum[region], op + (" [sgpr src]" if "SGPR" in cls else ""))):
            t[k][0] += 1; t[k][1] += w; t[k][2] += w * cyc

    tot_dyn = sum(v[1] for r in table.values() for v in r.values())
    tot_cyc = sum(v[2] for r in table.values() for v in r.values())
    iters = iters_ls * args.samples / args.lanestat_samples
    meas_cyc = args.kernel_ms * 1e-3 * args.clock_mhz * 1e6 * 1024 / iters
    print("kernel %s: %d VALU + %d other instructions (static)" % (args.kernel, n_valu, n_other))
    print("model: %.0f VALU wave-instructions and %.0f issue cycles per loop iteration (%.2f cycles each);" % (tot_dyn, tot_cyc, tot_cyc / tot_dyn))
    print("measured: %.2f ms x %.0f MHz x 1024 SIMDs / %.3g iterations = %.0f SIMD cycles per iteration -> the model's VALU issue time is %.0f %% of the kernel's time"
          % (args.kernel_ms, args.clock_mhz, iters, meas_cyc, 100.0 * tot_cyc / meas_cyc))
    def pair_cycles(v):
        return max(2.35 * (v[0] + v[1]), 4.2 * v[1]) + 4.3 * v[2] + 8.8 * v[3] + 16.3 * v[4]
    tot_pair = sum(pair_cycles(v) for v in pair.values())
    print("pair-aware model (an instruction of the 4.3-cycle kind overlaps with one of the 2.5-cycle kind; binary64 and transcendentals do not): %.0f cycles per iteration = %.0f %% of the kernel's time"
          % (tot_pair, 100.0 * tot_pair / meas_cyc))
    print()
    classes = sorted({c for r in table.values() for c in r}, key=lambda c: -sum(table[r][c][2] for r in table if c in table[r]))
    print("== by class (all regions)")
    print("%-34s %7s %9s %9s %7s %7s" % ("class", "static", "dyn/iter", "cyc/iter", "share", "cyc/op"))
    for c in classes:
        s = sum(table[r][c][0] for r in table if c in table[r]); d = sum(table[r][c][1] for r in table if c in table[r]); y = sum(table[r][c][2] for r in table if c in table[r])
        print("%-34s %7d %9.1f %9.1f %6.1f%% %7.2f" % (c, s, d, y, 100 * y / tot_cyc, y / d if d else 0))
    print("%-34s %7d %9.1f %9.1f %6.1f%%" % ("TOTAL", n_valu, tot_dyn, tot_cyc, 100.0))
    print()
    print("== by region")
    print("%-46s %7s %7s %9s %9s %7s %9s %6s  %s" % ("region", "w/iter", "static", "dyn/iter", "cyc/iter", "share", "pair-cyc", "lanes", "cycles by class"))
    regions = sorted(table, key=lambda r: -sum(v[2] for v in table[r].values()))
    rows = []
    for r in regions:
        s = sum(v[0] for v in table[r].values()); d = sum(v[1] for v in table[r].values()); y = sum(v[2] for v in table[r].values())
        top = sorted(table[r].items(), key=lambda kv: -kv[1][2])[:4]
        print("%-46s %7.3f %7d %9.1f %9.1f %6.1f%% %9.1f %6.1f  %s" % (r, W[key_of[r]], s, d, y, 100 * y / tot_cyc, pair_cycles(pair[r]), OCC[key_of[r]], ", ".join("%s %.0f" % (k, v[2]) for k, v in top)))
        for c, v in table[r].items():
            rows.append((r, c, W[key_of[r]], v[0], v[1], v[2], 100 * v[2] / tot_cyc, OCC[key_of[r]]))
    print()
    print("== the compare / select / min-max / convert / slow-int kind per region (cycles per iteration), and what full-rate forms would cost")
    slow_cls = ("compare", "select", "min/max/med3", "convert", "int / bit ops (slow forms)", "f32 add/mul, SGPR source", "move", "lane ops", "division fix-ups")
    print("%-46s" % "region" + "".join("%9s" % c[:8] for c in slow_cls) + "%9s" % "sum")
    for r in regions:
        vals = [table[r][c][2] if c in table[r] else 0.0 for c in slow_cls]
        if sum(vals) >= 5.0:
            print("%-46s" % r + "".join("%9.0f" % v for v in vals) + "%9.0f" % sum(vals))
    vals = [sum(table[r][c][2] for r in table if c in table[r]) for c in slow_cls]
    print("%-46s" % "TOTAL" + "".join("%9.0f" % v for v in vals) + "%9.0f  (%.1f %% of the VALU time)" % (sum(vals), 100 * sum(vals) / tot_cyc))
    print()
    print("== top opcode forms per region (dynamic count per iteration x cycles)")
    for r in regions[:14]:
        top = sorted(opsum[r].items(), key=lambda kv: -kv[1][2])[:args.ops]
        print("%-46s %s" % (r, "; ".join("%s %.0fx=%.0f" % (k, v[1], v[2]) for k, v in top)))
    if args.csv:
        os.makedirs(os.path.dirname(os.path.abspath(args.csv)), exist_ok=True)
        with open(args.csv, "w") as f:
            f.write("region,class,executions_per_iteration,static_instructions,dynamic_per_iteration,cycles_per_iteration,share_of_valu_time_pct,mean_active_lanes\n")
            for row in rows:
                f.write("%s,%s,%.4f,%d,%.2f,%.2f,%.3f,%.1f\n" % row)


if __name__ == "__main__":
    main()
