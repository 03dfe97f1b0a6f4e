"""The documents a reviewer reads name files; every file they name exists.  (README.md, INTEGRATION.md and the round-6 section
of profiles/README.md: a path in backticks ending in a source / data extension must resolve in the repository -- as written, or under
meshflow_amd/, meshflow_amd/csrc/, tests/, tests/golden/, tools/, oracle/ or profiles/; `{a,b}` and `*` are globbed.)  And the size caps the
round-5 review set: DESIGN.md <= 400 lines, README.md's status <= 40 lines."""
import glob
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROOTS = ['', 'meshflow_amd', os.path.join('meshflow_amd', 'csrc'), 'tests', os.path.join('tests', 'golden'), 'tools', 'oracle', 'profiles', 'include']
# generic patterns and files of the reference (not in this repository) that the documents name on purpose
ALLOWED = {'rNN*_traffic_rdreq.csv', 'rNN*_sq_warp.csv', 'rNN*_kernel_stats_<workload>.csv', 'meshflowstabilizer.py', 'motion_oracle.py/.c', 'mfs.py'}


def _named_files(text):
    for m in re.finditer(r'`([A-Za-z0-9_./*{},<>\-]+\.(?:py|sh|hip|h|c|md|json|csv|txt|npz))`', text):
        yield m.group(1)


def _exists(path):
    pattern = re.sub(r'\{[^}]*\}', '*', path)
    return any(glob.glob(os.path.join(REPO, root, pattern)) for root in ROOTS)


def test_every_file_the_documents_name_exists():
    missing = []
    for doc in ('DESIGN.md', 'README.md', 'INTEGRATION.md'):
        with open(os.path.join(REPO, doc)) as fh:
            text = fh.read()
        missing += [(doc, p) for p in _named_files(text) if p not in ALLOWED and '<' not in p and not _exists(p)]
    with open(os.path.join(REPO, 'profiles', 'README.md')) as fh:
        text = fh.read()
    round6 = text[text.index('## Round 6'):text.index('## Round 4')]
    missing += [('profiles/README.md (round 6)', p) for p in _named_files(round6) if p not in ALLOWED and '<' not in p and not _exists(p)]
    assert not missing, missing


def test_design_and_readme_stay_within_their_caps():
    with open(os.path.join(REPO, 'DESIGN.md')) as fh:
        design = fh.read().split('\n')
    assert len(design) <= 400, len(design)
    assert max(len(l) for l in design if not l.startswith('|')) <= 160          # (prose lines; table rows are one line each by syntax)
    with open(os.path.join(REPO, 'README.md')) as fh:
        readme = fh.read()
    status = readme[readme.index('## Status'):readme.index('Layout:')]
    assert len(status.strip().split('\n')) <= 40
