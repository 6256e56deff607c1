"""The generated asm blocks of the row kernel are committed; they must match their generator."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_fused_header_matches_generator():
    spec = importlib.util.spec_from_file_location(
        "gen_row_fused", os.path.join(ROOT, "scripts", "gen_row_fused.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    with open(gen.OUT) as f:
        assert f.read() == gen.render(), "run `python scripts/gen_row_fused.py` and rebuild"


def test_row_fused_blocks_respect_the_dpp_hazard_distance():
    """Inside every generated asm block no DPP source register (operand number) is written by
    either of the two preceding instructions (VALU write -> DPP read needs 2 wait states;
    s_nop N counts N + 1)."""
    spec = importlib.util.spec_from_file_location(
        "gen_row_fused", os.path.join(ROOT, "scripts", "gen_row_fused.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = gen.render()
    blocks, cur = [], None
    for line in text.splitlines():
        t = line.strip()
        if t.startswith("asm volatile("):
            cur = []
        elif cur is not None and t.startswith('"'):
            cur.append(t.strip('"').replace("\\n", ""))
        elif cur is not None and t.startswith(":"):
            blocks.append(cur)
            cur = None
    assert len(blocks) > 40
    checked = 0
    for blk in blocks:
        written = []          # per instruction slot: operand written (or None), wait states it provides
        for ins in blk:
            parts = ins.replace(",", " ").split()
            if parts[0] == "s_nop":
                written.append((None, int(parts[1]) + 1))
                continue
            dst = parts[1]
            if parts[0].endswith("_dpp"):
                src = parts[2]
                wait = 0
                for w, ws in reversed(written):
                    if wait >= 2:
                        break
                    assert w != src, f"DPP source {src} written {wait} wait states earlier in: {blk}"
                    wait += ws
                checked += 1
            written.append((dst, 1))
    assert checked > 300
