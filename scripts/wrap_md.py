"""Reflow the prose of a markdown file at WIDTH columns: a paragraph or list item any of whose lines is over-long (or was left
ragged by an earlier line-by-line wrap: a short line in its middle) is joined and wrapped again with its hanging indent.
Tables, code fences, indented code, headings and HTML are left alone.  Formatting only: the words are untouched (checked:
whitespace-normalised text before == after).  usage: wrap_md.py FILE..."""
import re
import sys
import textwrap

WIDTH = 124
BULLET = re.compile(r"^(\s*)((?:[-*+]|\d+\.)\s+)")


def keep(line):
    s = line.lstrip()
    return (not s) or s.startswith("|") or line.startswith("#") or line.startswith("    ") and not BULLET.match(line) and False


def reflow(item):
    """item: lines of one paragraph / list item (first line may carry the bullet)"""
    need = any(len(l) > WIDTH + 6 for l in item) or any(len(l) < 72 for l in item[:-1])
    if not need or len(item) == 0:
        return item
    m = BULLET.match(item[0])
    indent, bullet = (m.group(1), m.group(2)) if m else (re.match(r"^\s*", item[0]).group(0), "")
    body = " ".join(l.strip() for l in [item[0][len(indent) + len(bullet):]] + item[1:])
    return textwrap.wrap(body, WIDTH, initial_indent=indent + bullet, subsequent_indent=indent + " " * len(bullet),
                         break_long_words=False, break_on_hyphens=False)


def wrap_file(path):
    src = open(path).read()
    lines = src.split("\n")
    out, item, fence = [], [], False

    def flush():
        nonlocal item
        out.extend(reflow(item))
        item = []
    for line in lines:
        s = line.lstrip()
        if s.startswith("```"):
            flush()
            fence = not fence
            out.append(line)
            continue
        code = line.startswith("    ") and not item          # indented code / formulas: only outside a running item
        if fence or not s or s.startswith("|") or line.startswith("#") or s.startswith("<") or code:
            flush()
            out.append(line)
            continue
        if BULLET.match(line) and item:
            flush()
        item.append(line)
    flush()
    new = "\n".join(out)
    assert re.sub(r"\s+", " ", src) == re.sub(r"\s+", " ", new), path
    open(path, "w").write(new)


for p in sys.argv[1:]:
    wrap_file(p)
