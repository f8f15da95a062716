"""Per-kernel register / scratch / LDS usage of libgte_hip.so's code objects, from hipcc's -Rpass-analysis=kernel-resource-usage.

    python profiles/resource_usage.py [file.hip ...]      # default: every csrc/*.hip; prints a table, writes JSON with --json PATH

tests/test_abi_and_host.py uses parse() on the logs the Makefile leaves under csrc/_build/*.ru to assert that no kernel on the
step path spills (ScratchSize == 0)."""
import json
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gnn-tableextraction_amd", "csrc")
_KEYS = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
         "LDS Size [bytes/block]": "lds", "SGPRs": "sgprs"}


def parse(text: str):
    """-> {mangled kernel name: {vgprs, agprs, scratch, occupancy, lds, sgprs}}"""
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: (?:Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for k, name in _KEYS.items():
            m = re.search(re.escape(k) + r": (\d+)", line)
            if m:
                cur[name] = int(m.group(1))
    return out


def compile_usage(path: str):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++17", "-Rpass-analysis=kernel-resource-usage",
           "-c", path, "-o", "/dev/null"]
    r = subprocess.run(cmd, cwd=os.path.dirname(path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-2000:])
    return parse(r.stdout)


def demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), stdout=subprocess.PIPE, text=True)
        return dict(zip(names, r.stdout.splitlines()))
    except OSError:
        return {n: n for n in names}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    files = args or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    table = {}
    for f in files:
        table.update(compile_usage(os.path.abspath(f)))
    dm = demangle(list(table))
    for k, v in sorted(table.items(), key=lambda kv: dm[kv[0]]):
        flag = "  <-- SPILLS" if v.get("scratch", 0) else ""
        print(f"{v.get('vgprs', 0):4d} V {v.get('agprs', 0):4d} A {v.get('scratch', 0):5d} scratch {v.get('lds', 0):7d} LDS "
              f"occ {v.get('occupancy', 0)}  {dm[k][:150]}{flag}")
    if "--json" in sys.argv:
        json.dump({dm[k]: v for k, v in table.items()}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1, sort_keys=True)
