"""Re-wrap the over-long prose lines of a markdown file at 158 columns (tables, headings and code fences are left alone):
   python scripts/dev/wrap_md.py DESIGN.md"""
import sys
W = 158
path = sys.argv[1]
lines = open(path, encoding="utf8").read().split("\n")
out, fence, i = [], False, 0
special = lambda l: l.startswith(("|", "#", "```")) or not l.strip()
while i < len(lines):
    l = lines[i]
    if l.startswith("```"):
        fence = not fence
    if fence or special(l) or len(l) <= 160:
        out.append(l); i += 1; continue
    indent = len(l) - len(l.lstrip(" "))
    if l.lstrip().startswith(("* ", "- ")) or (l.lstrip()[:2].rstrip(".").isdigit() and l.lstrip()[1:3] in (". ",)):
        indent += 2 if l.lstrip()[0] in "*-" else 3
    cut = l.rfind(" ", 0, W)
    head, rest = l[:cut], l[cut + 1:]
    out.append(head)
    nxt = lines[i + 1] if i + 1 < len(lines) else ""
    cont = (not special(nxt)) and not nxt.lstrip().startswith(("* ", "- ")) and not (nxt.lstrip()[:1].isdigit() and nxt.lstrip()[1:3] == ". ") and not fence
    if cont:
        lines[i + 1] = " " * (len(nxt) - len(nxt.lstrip(" "))) + rest + " " + nxt.lstrip(" ")
    else:
        lines.insert(i + 1, " " * indent + rest)
    i += 1
open(path, "w", encoding="utf8").write("\n".join(out))
