"""Re-wrap the prose of a markdown file to a column limit (tables, code fences and headings untouched):
    python scripts/reflow_md.py DESIGN.md 120"""
import re
import sys
import textwrap

path, width = sys.argv[1], int(sys.argv[2])
out, para, fence = [], [], False


def flush():
    global para
    if not para:
        return
    if all(len(l) <= width for l in para):
        out.extend(para)
        para = []
        return
    first = para[0]
    m = re.match(r"^(\s*)([*-] |\d+\. )?", first)
    indent, bullet = m.group(1), m.group(2) or ""
    text = " ".join([first[len(indent) + len(bullet):].strip()] + [l.strip() for l in para[1:]])
    out.extend(textwrap.wrap(text, width=width, initial_indent=indent + bullet, subsequent_indent=indent + " " * len(bullet),
                             break_long_words=False, break_on_hyphens=False))
    para = []


for line in open(path, encoding="utf8").read().split("\n"):
    if line.startswith("```"):
        flush()
        fence = not fence
        out.append(line)
    elif fence or line.startswith("|") or line.startswith("#") or not line.strip():
        flush()
        out.append(line)
    elif re.match(r"^\s*([*-] |\d+\. )", line):
        flush()
        para = [line]
    else:
        para.append(line)
flush()
open(path, "w", encoding="utf8").write("\n".join(out))
