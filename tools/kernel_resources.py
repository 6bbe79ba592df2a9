#!/usr/bin/env python3
"""Registers, LDS and scratch of every gfx950 kernel in build/*.o (the objects `make` leaves), from the code-object metadata:
llvm-objcopy dumps the fat binary, clang-offload-bundler takes the gfx950 code object out, llvm-readelf prints its notes.
Exit status 1 when any kernel spills VGPRs to scratch: a spilling kernel is slower than its occupancy bound promises, and round 2
saw k_ovl_nei_fast<12|21> (4 spilled registers under __launch_bounds__(64, 6)) return WRONG values for a few strands in 10^7 when it
ran beside the walk kernel on a second stream, never alone -- scratch is not something to lean on here.
Usage: python tools/kernel_resources.py [-q] [objects...]"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], {}
    for ln in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(\S+)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "agpr_count" and cur.get("name"):      # first key of a kernel entry
            out.append(cur); cur = {}
        if k in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size", "agpr_count"):
            cur[k] = v
    if cur.get("name"):
        out.append(cur)
    return [k for k in out if "vgpr_count" in k]


def scratch_instructions(obj, mangled):
    """scratch_* / buffer_* (the two ways private memory is reached on gfx9) instructions in the code of one kernel of `obj`: a kernel whose metadata
    names a private segment but whose code has none of them holds SGPR spill slots that ended up in VGPR lanes -- it never touches memory for them."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--disassemble-symbols=" + mangled, co], check=True, capture_output=True, text=True).stdout
    n_ins = sum(1 for ln in dis.splitlines() if "//" in ln)
    assert n_ins > 50, "no disassembly of " + mangled
    return sum(1 for ln in dis.splitlines() if re.search(r"\b(scratch_|buffer_)", ln))


def demangle(sym):
    for tool in ("c++filt", LLVM + "/llvm-cxxfilt"):
        try:
            return subprocess.run([tool, sym], capture_output=True, text=True, check=True).stdout.strip().split("(")[0]
        except (OSError, subprocess.CalledProcessError):
            pass
    return sym


def main():
    args = [a for a in sys.argv[1:] if a != "-q"]
    quiet = "-q" in sys.argv[1:]
    objs = args or sorted(glob.glob(os.path.join(ROOT, "build", "*.o")))
    if not objs:
        print("no objects under build/ (run make first)")
        return 2
    bad = 0
    for o in objs:
        for k in kernels_of(o):
            name = demangle(k["name"])
            sp = int(k.get("vgpr_spill_count", 0))
            bad += sp > 0
            if not quiet or sp:
                print("%-18s %-44s vgpr %3s  sgpr %3s  lds %6s  scratch %4s  spilled vgprs %d%s" % (os.path.basename(o), name[:44], k["vgpr_count"], k.get("sgpr_count", "?"),
                      k.get("group_segment_fixed_size", "?"), k.get("private_segment_fixed_size", "?"), sp, "   <-- SPILLS" if sp else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
