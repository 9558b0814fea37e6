#!/usr/bin/env python3
"""Write `native_operator_manifest.json` next to libferrum_hip.so in the reference's manifest schema
(ferrum-types/src/native_operator.rs:39-58; validated fail-closed by ferrum-native-ops/src/resolver.rs:136-318: the file's
sha256 must match and `nm -g` must show every declared export).  backend is "hip" — the one-line enum extension
INTEGRATION.md §1 asks of the reference (its `cuda` branch insists on sm_xx capabilities, native_operator.rs:115-126)."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ferrum-infer-rs_amd", "lib", "libferrum_hip.so")
SRC_DIRS = [os.path.join(ROOT, "ferrum-infer-rs_amd", "csrc"), os.path.join(ROOT, "include")]


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def sha256_inputs():
    h = hashlib.sha256()
    for d in SRC_DIRS:
        for dirpath, _, files in sorted(os.walk(d)):
            for f in sorted(files):
                if f.endswith((".hip", ".cc", ".h", "Makefile")):
                    p = os.path.join(dirpath, f)
                    h.update(os.path.relpath(p, ROOT).encode())
                    h.update(sha256_file(p).encode())
    return h.hexdigest()


def main():
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    exports = sorted(line.split()[-1] for line in out.splitlines() if " T " in line and line.split()[-1].startswith("ferrum_"))
    try:
        rev = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
    except Exception:
        rev = "unknown"
    try:
        cc = subprocess.check_output(["/opt/rocm/bin/hipcc", "--version"], text=True).splitlines()[0].strip()
    except Exception:
        cc = "hipcc"
    inputs = sha256_inputs()
    manifest = {
        "schema_version": 1, "operator": "ferrum_hip_decode", "operator_abi_version": "1", "ferrum_native_abi_version": "1",
        "backend": "hip", "cuda_toolkit": None, "cuda_runtime_min": None, "compute_capabilities": ["gfx950"],
        "source_package": {"kind": "git", "revision": rev, "sha256": inputs},
        "inputs_sha256": inputs, "binary_sha256": sha256_file(LIB), "linkage": "dynamic", "exports": exports,
        "license_files": [],
        "build_summary": {"builder_sha": rev, "elapsed_ms": int((time.time() - os.path.getmtime(LIB)) * 0) , "nvcc_version": None,
                          "host_compiler": cc},
    }
    dst = os.path.join(os.path.dirname(LIB), "native_operator_manifest.json")
    json.dump(manifest, open(dst, "w"), indent=1)
    print(dst, len(exports), "exports")


if __name__ == "__main__":
    sys.exit(main())
