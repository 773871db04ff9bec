"""The node step of the timed packet-kernel instantiations as the compiler emits it: instruction counts by issue class.

    python3 tools/node_step_isa.py > profiles/r05_node_step_isa.txt      (runs hipcc -S on csrc/packetkernel.hip; no GPU needed)

The step is the straight-line code from the node's gathers to the branch-free stack tail (the basic blocks round the ten
v_min_f64 / v_max_f64 of the key sort).  Classes as measured by tools/micro/valu_issue.hip (profiles/r05_valu_ceiling.txt):
fast = 2.35 clocks per wave64 instruction per SIMD at even occupancies, 3.06 at three waves; slow = 4.3; trans = 8.2."""
import collections, os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_mul_legacy_f32", "v_addc_co_u32", "v_add_co_u32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32"}
def assembly(src, flags):
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "pk.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-ffp-contract=off"] + flags + ["-include", "cstring",
                               "-I" + os.path.join(REPO, "include"), "-S", "--cuda-device-only", os.path.join(REPO, "minimaloptix_amd", "csrc", src), "-o", asm],
                              stderr=subprocess.DEVNULL)
        return open(asm).read().split("\n")
# as the Makefile builds them: the 64-byte-node instantiations with LLVM's max-ilp scheduling (packetkernel.hip), the 128-byte-node ones with the default (packetkernel_n128.hip)
ASM = {"1": assembly("packetkernel.hip", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]), "0": assembly("packetkernel_n128.hip", [])}
for n64, label in (("1", "64-byte nodes (the benchmark's instantiation pt_packetkernel<false,true,false,false,true>)"), ("0", "128-byte nodes fetched by the ray's signs (<false,true,false,false,false>)")):
    name = "_ZN2pt12_GLOBAL__N_115pt_packetkernelILb0ELb1ELb0ELb0ELb%sEEEvNS_10LaunchArgsE" % n64
    txt = ASM[n64]
    start = [i for i, l in enumerate(txt) if l.startswith(name + ":")][0]
    end = [i for i, l in enumerate(txt) if i > start and l.startswith(".Lfunc_end")][0]
    body = txt[start:end]
    first = [i for i, l in enumerate(body) if "v_min_f64" in l][0]
    lo = first
    while not re.match(r"\s*global_load_dwordx4", body[lo]): lo -= 1
    while re.match(r"\s*(global_load_dwordx4|v_or_b32|v_lshlrev_b32|v_add_u32)", body[lo - 1]): lo -= 1
    hi = first
    while "s_branch" not in body[hi]: hi += 1
    c = collections.Counter()
    for l in body[lo:hi]:
        m = re.match(r"\s+([vsdg][a-z_0-9]+)", l)
        if m: c[re.sub(r"_e(32|64)$", "", m.group(1))] += 1
    v = {k: n for k, n in c.items() if k.startswith("v_")}
    fast = sum(n for k, n in v.items() if k in FAST); trans = sum(n for k, n in v.items() if k in TRANS); slow = sum(v.values()) - fast - trans
    print("== %s" % label)
    print("   vector %d (fast class %d, 4.3-clock class %d, transcendental %d), scalar %d, gathers %d, LDS %d" % (
        sum(v.values()), fast, slow, trans, sum(n for k, n in c.items() if k.startswith("s_")), c["global_load_dwordx4"], sum(n for k, n in c.items() if k.startswith("ds_"))))
    print("   issue clocks at three waves per SIMD: %.0f" % (fast * 3.06 + slow * 4.3 + trans * 8.2))
    print("   " + ", ".join("%s %d" % kv for kv in sorted(v.items(), key=lambda kv: -kv[1])))
