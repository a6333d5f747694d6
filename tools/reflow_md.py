"""Re-wrap the paragraphs and list items of a Markdown file at 118 columns (tables, headings, code blocks untouched).
    python tools/reflow_md.py DESIGN.md"""
import re, sys, textwrap
p = sys.argv[1]
src = open(p).read().split('\n')
blocks, cur, code = [], None, False
item = re.compile(r'^(\s*(?:[-*]|\d+\.)\s+)')
def flush():
    global cur
    if cur: blocks.append(cur)
    cur = None
for ln in src:
    if ln.startswith('```'):
        flush(); code = not code; blocks.append(('raw', ln)); continue
    if code or ln.strip() == '' or ln.startswith('#') or ln.lstrip().startswith('|'):
        flush(); blocks.append(('raw', ln)); continue
    m = item.match(ln)
    if m:
        flush(); cur = ['par', m.group(1), ' ' * len(m.group(1)), ln[len(m.group(1)):].strip()]
    elif cur is None:
        cur = ['par', '', '', ln.strip()]
    else:
        cur[3] += ('' if cur[3].endswith('::') else ' ') + ln.strip()
flush()
out = []
for b in blocks:
    if b[0] == 'raw':
        out.append(b[1]); continue
    out += textwrap.wrap(b[3], width=118, initial_indent=b[1], subsequent_indent=b[2], break_long_words=False, break_on_hyphens=False)
open(p, 'w').write('\n'.join(out).rstrip('\n') + '\n')
