"""Build libtokenhmr_hip.so of ANOTHER commit of this repository into build_ab/<name>/ (no GPU needed: hipcc cross-compiles gfx950), so that
scripts/ab_same_box.py can time it interleaved with the current build in one process on one box.

    python scripts/build_ab_lib.py <git-ref> <name> [-DNAME=VALUE ...]      e.g.  python scripts/build_ab_lib.py 447e554 r4
    python scripts/build_ab_lib.py WORKTREE p1a1 -DTHMR_S16_PUBLISH=1       (the working tree as it is, with extra compiler defines: A/B of a source-level choice)

The commit's sources are exported to a temporary directory (git archive: the working tree is not touched) and compiled with that commit's
own __graft_entry__.build(); only the shipped library is copied (build_ab/ is git-ignored and travels to the GPU box with gpurun).
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) < 3 or any(not a.startswith("-D") for a in sys.argv[3:]):
        sys.exit(__doc__)
    ref, name, defines = sys.argv[1], sys.argv[2], sys.argv[3:]
    dst = os.path.join(ROOT, "build_ab", name)
    os.makedirs(dst, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        if ref == "WORKTREE":
            for d in ("tokenhmr_amd", "include"):
                shutil.copytree(os.path.join(ROOT, d), os.path.join(tmp, d), ignore=shutil.ignore_patterns("lib", "__pycache__"))
            shutil.copy(os.path.join(ROOT, "__graft_entry__.py"), tmp)
        else:
            tar = subprocess.run(["git", "-C", ROOT, "archive", ref, "tokenhmr_amd", "include", "__graft_entry__.py"], check=True, stdout=subprocess.PIPE).stdout
            subprocess.run(["tar", "-x", "-C", tmp], input=tar, check=True)
        subprocess.run([sys.executable, "-c", f"import __graft_entry__ as g; g.FLAGS += {defines!r}; g.build(experiments=False)"], cwd=tmp, check=True)
        shutil.copy(os.path.join(tmp, "tokenhmr_amd", "lib", "libtokenhmr_hip.so"), os.path.join(dst, "libtokenhmr_hip.so"))
    sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD" if ref == "WORKTREE" else ref], check=True, stdout=subprocess.PIPE, text=True).stdout.strip()
    with open(os.path.join(dst, "SOURCE"), "w") as f:
        f.write(f"{sha} ({ref}) {' '.join(defines)}\n")
    print(f"build_ab/{name}/libtokenhmr_hip.so <- {sha}")


if __name__ == "__main__":
    main()
