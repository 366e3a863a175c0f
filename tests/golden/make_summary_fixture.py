"""Extract the reference's stored ``model.summary()`` printout into a JSON fixture.

Source: /root/reference/notebooks/Train/Train_tests.ipynb (cell output, raw JSON lines 436-577) --
the only reference-owned golden for the hot path.  The fixture is DATA
(an expected output the reference recorded): layer name, class, output shape, parameter count and
inbound layers per row, plus the three totals.  Run here (the reference is not on the GPU box):

    python tests/golden/make_summary_fixture.py
"""
import json
import os
import re

NB = '/root/reference/notebooks/Train/Train_tests.ipynb'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'model_summary.json')


def main():
    nb = json.load(open(NB))
    text = None
    for cell in nb['cells']:
        for out in cell.get('outputs', []):
            t = ''.join(out.get('text', []))
            if 'Model: "unet"' in t and 'Total params' in t:
                text = t
    assert text is not None, 'summary output not found'
    rows = []
    cur = None
    for line in text.splitlines():
        m = re.match(r'^(\S+) \((\S+?)\)?\s+(\[?\(None,[^)]*\))\s+(\d+)\s*(\S.*)?$', line)
        if m:
            name, cls, shape, params, conn = m.groups()
            dims = [None if d.strip() == 'None' else int(d) for d in shape.strip('[]()').split(',') if d.strip()]
            cur = dict(name=name, type_prefix=cls, shape=dims, params=int(params),
                       inputs=[conn.strip().split('[')[0]] if conn and conn.strip() else [])
            rows.append(cur)
        elif cur is not None and re.match(r'^\s{40,}\S+\[\d+\]\[\d+\]', line):
            cur['inputs'].append(line.strip().split('[')[0])
    tot = {k: int(re.search(k + r' params: ([\d,]+)', text).group(1).replace(',', ''))
           for k in ('Total', 'Trainable', 'Non-trainable')}
    fixture = dict(source='notebooks/Train/Train_tests.ipynb:436-577 (stored cell output)',
                   config=dict(DIM=[128, 128], FILTERS=32, DEPTH=4, BATCH_NORMALISATION=True, BN_FIRST=False,
                               ACTIVATION='relu', MASK_CLASSES=2, IMG_CHANNELS=1, M_POOL=[2, 2], F_SIZE=[3, 3]),
                   rows=rows, totals=tot)
    json.dump(fixture, open(OUT, 'w'), indent=1)
    print('wrote', OUT, len(rows), 'rows', tot)


if __name__ == '__main__':
    main()
