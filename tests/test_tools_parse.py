"""tools/ holds the scripts the docs' numbers come from; most run only on the GPU box. Here: they at least parse (bash -n, ast)."""
import ast
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shell_scripts_parse():
    scripts = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert len(scripts) >= 10
    for f in scripts:
        p = subprocess.run(["bash", "-n", f], capture_output=True, text=True)
        assert p.returncode == 0, (f, p.stderr)


def test_python_tools_parse():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    assert len(files) >= 20
    for f in files:
        with open(f) as fh:
            ast.parse(fh.read(), filename=f)
