"""TEST INFRASTRUCTURE: the committed recipe that COMPILES the reference into ``oracle/_ref/`` (git-ignored; it travels to the
GPU box with the snapshot like this repo's own built ``.so``), so that the GPU node can run the REFERENCE ITSELF -- as the
CPU baseline of bench.py (``cpu_baseline.kind == "reference"``), in tests/test_oracle_vs_reference.py, and as the four phase
functions that tests/test_gpu_dropin.py drives over the HIP runner.

    python -m oracle.build_ref            (also run by __graft_entry__.build() whenever /root/reference is mounted)

The reference is Python, so "compiled" means CPython bytecode.  No source text is copied anywhere:
  * every module under /root/reference/src is byte-compiled from where it lies into a sourceless tree
    ``oracle/_ref/src/**/<module>.pyc`` (``py_compile``; the import system loads ``X.pyc`` next to a missing ``X.py``);
  * for the files whose modules cannot be imported here (they import torchvision / omegaconf / cv2 at module level) the
    loader takes single function / class definitions by name -- their code objects, compiled from the unmodified AST, are
    marshalled into ``oracle/_ref/defs/<path>.marshal`` as {name: code};
  * ``MANIFEST.json`` names the interpreter (bytecode is version-bound) and a SHA-256 over the compiled sources, so a stale
    or foreign ``_ref`` is refused instead of half-working.
Only oracle/reference_loader.py reads ``oracle/_ref``; nothing in the product package may (tests/test_cabi.py greps for it).
"""
import ast
import hashlib
import json
import marshal
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE_ROOT = "/root/reference"
REF_DIR = os.path.join(HERE, "_ref")

# files reference_loader._extract() takes definitions from (the module itself is not importable in this image)
DEF_FILES = [
    "src/core/generation_utils.py", "src/core/generation_phases.py", "src/data/image/transforms/side_resize.py",
    "src/data/image/transforms/divisible_crop.py", "src/data/image/transforms/na_resize.py", "src/utils/color_fix.py",
    "src/common/seed.py", "src/optimization/performance.py",
]


def interpreter_tag() -> str:
    return "cpython-%d.%d" % sys.version_info[:2]


def build(source_root: str = SOURCE_ROOT, out_dir: str = REF_DIR, verbose: bool = True) -> str:
    src_top = os.path.join(source_root, "src")
    if not os.path.isdir(src_top):
        raise FileNotFoundError(f"{src_top}: the reference checkout is not mounted here")
    tmp = out_dir + ".tmp"
    shutil.rmtree(tmp, ignore_errors=True)
    h = hashlib.sha256()
    n_mod = 0
    for dirpath, dirnames, filenames in os.walk(src_top):
        dirnames[:] = sorted(d for d in dirnames if d != "__pycache__")
        rel_dir = os.path.relpath(dirpath, source_root)
        for fn in sorted(filenames):
            if not fn.endswith(".py"):
                continue
            sp = os.path.join(dirpath, fn)
            with open(sp, "rb") as f:
                h.update(os.path.join(rel_dir, fn).encode() + b"\0" + f.read())
            dest = os.path.join(tmp, rel_dir, fn + "c")
            os.makedirs(os.path.dirname(dest), exist_ok=True)
            # dfile: the name tracebacks show (relative to the reference root; the file is not there on the GPU box)
            py_compile.compile(sp, cfile=dest, dfile=os.path.join("<reference>", rel_dir, fn), doraise=True, optimize=0)
            n_mod += 1
    n_defs = 0
    for rel in DEF_FILES:
        sp = os.path.join(source_root, rel)
        tree = ast.parse(open(sp).read())
        table = {}
        for node in tree.body:
            if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
                code = compile(ast.Module(body=[node], type_ignores=[]), os.path.join("<reference>", rel), "exec")
                table[node.name] = code
        dest = os.path.join(tmp, "defs", rel.replace("/", "__") + ".marshal")
        os.makedirs(os.path.dirname(dest), exist_ok=True)
        with open(dest, "wb") as f:
            marshal.dump(table, f)
        n_defs += len(table)
    with open(os.path.join(tmp, "MANIFEST.json"), "w") as f:
        json.dump({"interpreter": interpreter_tag(), "sources_sha256": h.hexdigest(), "modules": n_mod, "definitions": n_defs,
                   "recipe": "oracle/build_ref.py", "compiled_from": source_root}, f, indent=1)
    shutil.rmtree(out_dir, ignore_errors=True)
    os.rename(tmp, out_dir)
    if verbose:
        print(f"[oracle/_ref] {n_mod} reference modules byte-compiled, {n_defs} definitions marshalled -> {out_dir}")
    return out_dir


if __name__ == "__main__":
    build()
