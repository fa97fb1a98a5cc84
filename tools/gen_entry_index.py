"""every `extern "C"` symbol of include/trl_hip.h grouped by the header section ("/* --- title ---")
that documents it.  `python tools/gen_entry_index.py` prints the markdown table."""
import os
import re

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(HERE, "include", "trl_hip.h")).read()
sections, cur = [("library", [])], None
pos = 0
for m in re.finditer(r"/\*\s*---\s*([^\n]+)|\b(trl_[a-z0-9_]+)\s*\(", src):
    if m.group(1):
        title = re.sub(r"\s*-{3,}.*$", "", m.group(1)).strip()
        title = re.sub(r"\s*\*/\s*$", "", title)
        sections.append((title, []))
    else:
        # skip names that only occur inside comments
        line_start = src.rfind("\n", 0, m.start()) + 1
        before = src[:m.start()]
        if before.count("/*") > before.count("*/"):
            continue
        if m.group(2) not in [n for _, ns in sections for n in ns]:
            sections[-1][1].append(m.group(2))
print("| header section | entry points |\n|---|---|")
for title, names in sections:
    if names:
        print("| %s | %s |" % (title.split(":")[0] + (":" + title.split(":", 1)[1] if ":" in title else ""), ", ".join("`%s`" % n for n in names)))
print("\n%d entry points" % sum(len(n) for _, n in sections))
