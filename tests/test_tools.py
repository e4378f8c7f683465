"""The development scripts under tools/ are not part of the library, but the measurement pipeline (profiles/) is built from
them: keep them at least syntactically alive ("none run by a test")."""
import glob
import os
import py_compile
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_tools_compile(tmp_path):
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "numlab", "*.py")))
    assert files
    for f in files:
        py_compile.compile(f, cfile=str(tmp_path / (os.path.basename(f) + "c")), doraise=True)


def test_shell_tools_parse():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert files
    for f in files:
        subprocess.run(["bash", "-n", f], check=True)


def test_tools_readme_lists_every_script():
    readme = open(os.path.join(ROOT, "tools", "README.md")).read()
    for f in glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.sh")):
        name = os.path.basename(f)
        if name.startswith("run"):      # scratch launchers of one session
            continue
        assert name in readme, f"tools/README.md does not mention {name}"
