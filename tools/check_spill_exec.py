#!/usr/bin/env python
"""Build-time guard against a register-allocator defect of the ROCm 7.2 LLVM on gfx950 ("the order-dependent gradient"):
in a kernel that spills both SGPRs and VGPRs, VGPR spill stores can land in the PROLOGUE of a reconvergence block -- after an SGPR spill
(v_writelane) and BEFORE the `s_or_b64 exec, exec, sN` that re-enables the lanes of the other side of the branch.  The store then runs with
those lanes masked off; the matching reload runs under the full mask and hands them whatever the scratch arena held (zero in a fresh process,
stale data of earlier kernels otherwise): a result that depends on the history of the process.

The check disassembles every gfx950 code object of the library and reports each `s_or_b64 exec, exec, s[..]` that (a) starts a basic block or
follows only SGPR-spill / spill-store instructions from the start of one and (b) has a `scratch_store` between the block start and itself.
usage: check_spill_exec.py <lib.so | dir of .s files> ; exit status 1 when a kernel is hit."""
import os, re, shutil, struct, subprocess, sys, tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


class GuardError(RuntimeError):
    """The guard could not LOOK (no disassembler, no code object, objdump failed, nothing parsed): never the same thing as "0 hits"."""


def find_objdump():
    """llvm-objdump of the toolchain that built the library: $ROCM_PATH, the directory tree hipcc lives in, /opt/rocm, then PATH."""
    roots = [os.environ.get("ROCM_PATH"), os.environ.get("HIP_PATH")]
    hipcc = shutil.which("hipcc")
    if hipcc:
        roots.append(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))))
    roots.append("/opt/rocm")
    for r in roots:
        if r:
            for rel in ("lib/llvm/bin/llvm-objdump", "llvm/bin/llvm-objdump", "bin/llvm-objdump"):
                c = os.path.join(r, rel)
                if os.path.isfile(c) and os.access(c, os.X_OK):
                    return c
    c = shutil.which("llvm-objdump")
    if c:
        return c
    raise GuardError("check_spill_exec: no llvm-objdump found (ROCM_PATH, hipcc's tree, /opt/rocm, PATH)")


def code_objects(path):
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            break
        n, = struct.unpack_from("<Q", data, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx" in triple and size > 0:
                out.append(data[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def scan_disassembly(text):
    """text of llvm-objdump -d: returns [(kernel, address of the s_or, [addresses of the stores])]"""
    lines = text.split("\n")
    ins = []                     # (addr, mnemonic+operands, kernel)
    kern = None
    for l in lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:", l)
        if m:
            kern = m.group(1); continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            ins.append((int(m.group(2), 16), m.group(1).strip(), kern))
    targets = set()
    for a, t, k in ins:
        m = re.match(r"s_c?branch\w*\s+(-?\d+)", t)
        if m:
            targets.add(a + 4 + 4 * int(m.group(1)))         # SOPP branch: target = pc + 4 + simm16 * 4
    hits = []
    for i, (a, t, k) in enumerate(ins):
        if not re.match(r"s_or_b64 exec, exec, s\[", t):
            continue
        stores, j = [], i - 1
        while j >= 0:
            aj, tj, kj = ins[j]
            if tj.startswith("scratch_store") or tj.startswith("buffer_store") and "offen" not in tj and "s[" in tj and "Spill" in tj:
                stores.append(aj)
            elif not (tj.startswith("v_writelane_b32") or tj.startswith("s_nop") or tj.startswith("s_waitcnt") or tj.startswith("s_mov_b")):
                break
            if aj in targets:                                  # reached the start of the block: everything walked over is its prologue
                if stores:
                    hits.append((k, a, stores))
                break
            j -= 1
    return only_store_sites(hits, [(a, t, k) for a, t, k in ins if t.startswith("scratch_store")], lambda x: x)


def only_store_sites(hits, all_stores, key):
    """A store in front of the exec restore is the defect only when the slot has NO other store site in the kernel: a value defined inside the
    divergent region is legitimately stored there under the region's mask (its other lanes were stored where THEY defined it); a value that lived
    in a register until this point and is spilled here for the first time loses the lanes that are masked off."""
    def slot(t):
        m = re.search(r"offset:(\d+)", t)
        return int(m.group(1)) if m else 0
    sites = {}
    for a, t, k in all_stores:
        w = re.match(r"scratch_store_dwordx(\d)", t)
        for d in range(int(w.group(1)) if w else 1):
            sites.setdefault((k, slot(t) + 4 * d), set()).add(a)
    text_of = {a: t for a, t, k in all_stores}
    out = []
    for k, a, stores in hits:
        lone = []
        here = set(stores)
        for s_ in stores:
            t = text_of[s_]
            w = re.match(r"scratch_store_dwordx(\d)", t)
            if any(sites[(k, slot(t) + 4 * d)] <= here for d in range(int(w.group(1)) if w else 1)):
                lone.append(s_)
        if lone:
            out.append((k, a, lone))
    return out


def scan_asm(text):
    """text of a compiler .s file (labels present)"""
    L, hits, kern = text.split("\n"), [], None
    all_stores = []
    for i, l in enumerate(L):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
        if l.strip().startswith("scratch_store"):
            all_stores.append((i + 1, l.strip(), kern))
    kern = None
    for i, l in enumerate(L):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
        if re.match(r"\.LBB\d+_\d+:", l):
            j, stores = i + 1, []
            while j < len(L):
                t = L[j].strip()
                if t.startswith("scratch_store"):
                    stores.append(j + 1)
                elif re.match(r"s_or_b64 exec, exec", t):
                    if stores:
                        hits.append((kern, j + 1, stores))
                    break
                elif not (t.startswith("v_writelane_b32") or t.startswith("s_nop") or t.startswith("s_waitcnt") or t.startswith("s_mov_b") or t.startswith(";") or t == ""):
                    break
                j += 1
    return only_store_sites(hits, all_stores, lambda x: x)


def _scan_code_object(co):
    objdump = find_objdump()
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        r = subprocess.run([objdump, "-d", f.name], capture_output=True, text=True)
    if r.returncode != 0:
        raise GuardError("check_spill_exec: %s -d failed (rc %d): %s" % (objdump, r.returncode, r.stderr.strip()[:400]))
    # a code object that holds kernels must yield kernel symbols and instructions; an empty parse means the output format changed under the scanner
    n_sym = len(re.findall(r"^[0-9a-f]+ <.+>:", r.stdout, re.M))
    n_ins = len(re.findall(r"^\s+\S.*?//\s*[0-9A-Fa-f]+:", r.stdout, re.M))
    if n_sym == 0 or n_ins == 0:
        raise GuardError("check_spill_exec: nothing parsed from the disassembly of a %d-byte code object (%d symbols, %d instructions)" % (len(co), n_sym, n_ins))
    return scan_disassembly(r.stdout)


def check_library(path, verbose=True, min_code_objects=1):
    """Raises GuardError when the library cannot be inspected (fails CLOSED): no gfx code object found in `path` (e.g. a compressed 'CCOB' bundle
    this reader does not unpack), fewer than `min_code_objects`, a failing objdump, or a disassembly without kernel symbols."""
    from concurrent.futures import ThreadPoolExecutor
    cos = code_objects(path)
    if len(cos) < max(1, min_code_objects):
        raw = open(path, "rb").read()
        hint = " (compressed offload bundle 'CCOB' present: build with --no-offload-compress)" if b"CCOB" in raw else ""
        raise GuardError("check_spill_exec: %s holds %d gfx code object(s), expected >= %d%s" % (path, len(cos), max(1, min_code_objects), hint))
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(cos)))) as ex:
        res = list(ex.map(_scan_code_object, cos))
    hits = [(n,) + x for n, h in enumerate(res) for x in h]
    if verbose:
        for n, k, a, st in hits:
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:160]
            print("code object %d: %s: %d spill store(s) before the exec restore at 0x%x" % (n, name, len(st), a))
    return hits


if __name__ == "__main__":
    p = sys.argv[1]
    if p.endswith(".s"):
        h = scan_asm(open(p).read())
        for k, ln, st in h:
            print("%s: line %d: %d spill store(s) before the exec restore" % (k, ln, len(st)))
    else:
        h = check_library(p)
    print("%d hit(s)" % len(h))
    sys.exit(1 if h else 0)
