"""Every run-time switch of the product -- `env_int("YL_...")` in the native sources, `os.environ.get("YL_...")` in the package -- is
listed in INTEGRATION.md (section 5); development constants (`tune_int`) are not run-time switches and are not required there."""
import pathlib
import re

ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_runtime_knobs_are_listed_in_integration_md():
    doc = (ROOT / "INTEGRATION.md").read_text()
    native = set()
    for f in list((ROOT / "yolo-lite_amd" / "csrc").glob("*.hip")) + list((ROOT / "yolo-lite_amd" / "csrc").glob("*.h")):
        native |= set(re.findall(r'env_int\("(YL_[A-Z0-9_]+)"', f.read_text()))
    host = set()
    for f in (ROOT / "yolo-lite_amd").rglob("*.py"):
        host |= set(re.findall(r'environ(?:\.get|\.setdefault)?\(\s*"(YL_[A-Z0-9_]+)"', f.read_text()))
        host |= set(re.findall(r'environ\[\s*"(YL_[A-Z0-9_]+)"\s*\]', f.read_text()))
    assert len(native) >= 20 and len(host) >= 5
    missing = sorted(k for k in native | host if k not in doc)
    assert not missing, f"not documented in INTEGRATION.md: {missing}"
