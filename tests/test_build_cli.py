"""fcc_qp_amd/build.py --add-fused-shape / --add-coop-shape: the one-line way to get a specialised kernel for another
robot's shape.  Only the list handling is exercised here; the build itself is __graft_entry__.build()."""
import importlib.util
import os
import shutil


def _load():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fcc_qp_amd_build_cli", os.path.join(root, "fcc_qp_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_add_shape_appends_once(tmp_path):
    B = _load()
    for name in ("fused_shapes.def", "coop_shapes.def"):
        shutil.copy(os.path.join(B.CSRC, name), tmp_path / name)
    B.CSRC = str(tmp_path)
    assert B.add_shape("fused", (60, 38, 12, 38)) is False          # the walking shape is listed
    assert B.add_shape("fused", (44, 30, 6, 30)) is True
    assert B.add_shape("fused", ("44", "30", "6", "30")) is False   # ... and only once
    text = (tmp_path / "fused_shapes.def").read_text()
    assert text.count("FCCQP_FUSED_SHAPE(44, 30, 6, 30)") == 1 and text.endswith("\n")
    assert B.add_shape("coop", (114, 60, 24, 66, 8)) is False
    assert B.add_shape("coop", (100, 50, 12, 60, 8)) is True
    assert "FCCQP_COOP_SHAPE(100, 50, 12, 60, 8)" in (tmp_path / "coop_shapes.def").read_text()
