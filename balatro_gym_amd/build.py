"""Build libbalatro_mi355x.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m balatro_gym_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
-ffp-contract=off: the reference's float64 expressions must not be fused into FMAs (bit-exact rewards/scores).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "bg_lib.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("bg_device.h", "bg_step.h", "bg_tables.h", "bg_ops.h", "bg_sim.h", "bg_engine.h", "bg_engine3.h")] + [
    os.path.join(os.path.dirname(HERE), "include", "balatro_mi355x.h")]
LIB = os.path.join(HERE, "libbalatro_mi355x.so")
ARCH = "gfx950"


def hipcc_path() -> str:
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: the MI355X kernels cannot be built")
    return p


FLAGS = ["-O3", f"--offload-arch={ARCH}", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value"]


def _sources_digest() -> str:
    """12 hex digits over every source the library is compiled from and the compiler flags: the box-independent part of the signature."""
    import hashlib
    h = hashlib.sha256()
    for d in DEPS:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:12]


_compiler_digest_cache: list = []


def _compiler_digest() -> str:
    """4 hex digits over hipcc's version lines (one `hipcc --version` per process)."""
    import hashlib
    if not _compiler_digest_cache:
        c = hashlib.sha256()
        try:
            ver = subprocess.run([hipcc_path(), "--version"], capture_output=True, text=True, timeout=60).stdout
            c.update("\n".join(l for l in ver.splitlines() if "version" in l.lower()).encode())
        except Exception:
            c.update(b"hipcc-unknown")
        _compiler_digest_cache.append(c.hexdigest()[:4])
    return _compiler_digest_cache[0]


def source_signature() -> str:
    """What identifies the DEVICE CODE of a build, reproducibly: 16 hex digits = sha256 prefix (12) over every source the library is
    compiled from and the compiler flags, then sha256 prefix (4) of the compiler's version line.  Two builds of unchanged sources with the
    same hipcc carry the same signature (the bytes of the code object do not: rebuilding unchanged sources gave three different .hip_fatbin
    hashes), so a committed PMC traffic measurement (profiles/*_hbm_traffic.json) stays attached to the code it was taken on across
    rebuilds.  The first 12 digits do not depend on the box (`sources_part`): a library that travelled to a box with another ROCm still
    says which sources it was built from."""
    return _sources_digest() + _compiler_digest()


def sources_part(signature: str) -> str:
    """The compiler-independent part of a build signature."""
    return signature[:12]


def library_signature(path: str = LIB) -> str | None:
    """The signature a built library carries (the string bg_build_signature() returns), read from the file: the 16 hex digits behind the marker
    the library stores in front of it.  None for a library without one (an ad-hoc build says "unsigned")."""
    import re
    try:
        with open(path, "rb") as f:
            m = re.search(rb"bgsig:([0-9a-f]{16})\0", f.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def needs_build() -> bool:
    """No library, or sources newer than it -- unless the library SAYS it was built from exactly these sources: a snapshot that does not keep
    modification times (a copy, a fresh checkout beside a travelled .so) must not set off a three-minute rebuild of unchanged code, and must not
    replace the library the committed measurements belong to.  The library's own signature string is compared (not "these 12 digits occur
    somewhere in the file"); a library of the same sources built by ANOTHER compiler is kept -- it travelled from the build container -- and said so on stderr."""
    if not os.path.exists(LIB):
        return True
    if not any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in DEPS):
        return False
    have = library_signature()
    if have is None or sources_part(have) != _sources_digest():
        return True
    try:
        if have != source_signature():
            print(f"balatro_gym_amd.build: {LIB} was built from the current sources by another compiler (signature {have}, here {source_signature()}): kept",
                  file=sys.stderr)
    except Exception:
        pass
    return False


def build(force: bool = False, verbose: bool = False, out: str | None = None) -> str:
    if not force and not out and not needs_build():
        return LIB
    cmd = [hipcc_path()] + FLAGS + [f'-DBG_BUILD_SIGNATURE="{source_signature()}"', "-o", out or LIB, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out or LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
