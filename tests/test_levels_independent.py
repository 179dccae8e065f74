"""The oracle's level data (oracle/levels_oracle.h) and the product's (include/sgk_levels.h) are two transcriptions of the same
levels, written independently; this test is where they meet. Every level's shape and ASCII art, every character constant, reward
constant, probability threshold and switch default, the observation value and the colour of every printable character must agree,
field by field, under the default switches -- and the oracle must not read the product's header any more."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC = os.path.join(ROOT, "oracle")


def _dump(which):
    subprocess.check_call(["make", "-s", "-C", ORC, "levels"])
    lib = ctypes.CDLL(os.path.join(ORC, "liblevels_%s.so" % which))
    lib.levels_dump.restype = ctypes.c_size_t
    lib.levels_dump.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    buf = ctypes.create_string_buffer(1 << 18)
    need = lib.levels_dump(buf, len(buf))
    assert need <= len(buf)
    return buf.value.decode().splitlines()


def test_the_two_level_tables_agree_field_by_field():
    oracle, product = _dump("oracle"), _dump("product")
    assert len(oracle) > 200  # ~100 named constants + ten levels x (shape + art rows + characters)
    fields = lambda lines: {ln.split(" = ")[0] if " = " in ln else ln.split(":")[0]: ln for ln in lines}  # noqa: E731
    fo, fp = fields(oracle), fields(product)
    assert sorted(fo) == sorted(fp), "the tables do not even list the same fields: %s" % sorted(set(fo) ^ set(fp))
    wrong = [(fo[k], fp[k]) for k in fo if fo[k] != fp[k]]
    assert not wrong, "oracle vs product:\n" + "\n".join("  %s   |   %s" % w for w in wrong)
    assert oracle == product  # (same order too: one dump routine)
    # all ten levels are in it, with their art
    for env in range(10):
        assert any(ln.startswith("level %d art[0]" % env) for ln in oracle)


def test_the_oracle_does_not_read_the_products_level_table():
    src = open(os.path.join(ORC, "sgk_oracle.c")).read()
    assert not re.search(r'#\s*include\s+"[^"]*sgk_levels\.h"', src)
    assert re.search(r'#\s*include\s+"levels_oracle\.h"', src)
    hdr = open(os.path.join(ORC, "levels_oracle.h")).read()
    assert "#include" not in hdr.replace("#include <", "")  # the oracle's header pulls in nothing of the product's
    mk = open(os.path.join(ORC, "Makefile")).read()
    assert re.search(r"^\$\(OUT\): sgk_oracle\.c levels_oracle\.h$", mk, re.M)


def test_art_rows_have_the_declared_width_and_one_agent():
    for which in ("oracle", "product"):
        shapes, rows = {}, {}
        for ln in _dump(which):
            m = re.match(r"level (\d+) shape = (\d+) x (\d+)", ln)
            if m:
                shapes[int(m.group(1))] = (int(m.group(2)), int(m.group(3)))
            m = re.match(r'level (\d+) art\[(\d+)\] = "(.*)" \((\d+) chars\)', ln)
            if m:
                rows.setdefault(int(m.group(1)), []).append(m.group(3))
                assert len(m.group(3)) == int(m.group(4))
        for env, (h, w) in shapes.items():
            assert len(rows[env]) == h and all(len(r) == w for r in rows[env]), (which, env)
            assert sum(r.count("A") for r in rows[env]) == 1 and h * w <= 64, (which, env)
