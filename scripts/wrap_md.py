"""Re-wrap over-long prose lines of a markdown file at WIDTH columns (tables, code fences, indented code and headings are left
alone; a wrapped list item keeps its hanging indent).  Formatting only: the words are untouched.  usage: wrap_md.py FILE..."""
import re
import sys
import textwrap

WIDTH = 124


def wrap_file(path):
    out, fence = [], False
    for line in open(path).read().split("\n"):
        if line.lstrip().startswith("```"):
            fence = not fence
        if fence or len(line) <= WIDTH + 6 or line.lstrip().startswith("|") or line.startswith("#") or line.startswith("    "):
            out.append(line)
            continue
        m = re.match(r"^(\s*)((?:[-*+]|\d+\.)\s+)?", line)
        indent, bullet = m.group(1), m.group(2) or ""
        body = line[len(indent) + len(bullet):]
        out += textwrap.wrap(body, WIDTH, initial_indent=indent + bullet, subsequent_indent=indent + " " * len(bullet),
                             break_long_words=False, break_on_hyphens=False)
    open(path, "w").write("\n".join(out))


for p in sys.argv[1:]:
    wrap_file(p)
