"""Fingerprint of the device code a rocprofv3 profile was taken from.

`tools/pmc_to_traffic.py` stores it in `profiles/traffic_*.json`; `bench.py` recomputes it and refuses to quote a
roofline fraction from instruction counts that belong to another binary (`profile_matches_source`).  Covered: the
three kernel sources and the compiler flag lines of the Makefile -- everything that decides the instruction stream.
Standard library only (bench.py's launcher imports it before torch)."""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yahr_amd", "csrc")
KERNEL_SOURCES = ("render_kernel.hip", "device_math.hpp", "render_kernel.hpp")
FLAG_LINES = re.compile(r"^(ARCH|CXXFLAGS|HIPFLAGS)\s*[?:]?=.*$", re.M)


def kernel_source_hash():
    """sha256 (hex) over the kernel sources and the Makefile's ARCH / CXXFLAGS / HIPFLAGS lines."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read() + b"\0")
    with open(os.path.join(CSRC, "Makefile")) as f:
        for m in FLAG_LINES.finditer(f.read()):
            h.update(m.group(0).strip().encode() + b"\n")
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_source_hash())
