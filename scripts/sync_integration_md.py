"""Keeps INTEGRATION.md section 2 the verbatim text of integration/hipapi.h (tests/test_capi_exports.py checks it).  usage: python scripts/sync_integration_md.py"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
md_path = os.path.join(ROOT, "INTEGRATION.md")
md = open(md_path).read()
shim = open(os.path.join(ROOT, "integration", "hipapi.h")).read()
m = re.search(r"```cpp\n/\*\n  hipapi\.h -- .*?#endif // HIPAPI_H\n```", md, re.S)
assert m, "section 2's code block not found"
md = md[:m.start()] + "```cpp\n" + shim + "```" + md[m.end():]
open(md_path, "w").write(md)
print("INTEGRATION.md section 2 synced with integration/hipapi.h")
