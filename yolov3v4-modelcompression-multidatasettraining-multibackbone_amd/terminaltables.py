"""Minimal stand-in for the ``terminaltables`` package (not installed in this image): the prune scripts of the reference
(slim_prune.py, layer_prune.py, utils/prune_utils.py:2) only build ``AsciiTable(rows).table`` strings for printing."""


class AsciiTable:
    def __init__(self, table_data, title=None):
        self.table_data = [[str(c) for c in row] for row in table_data]
        self.title = title
        self.inner_heading_row_border = True

    @property
    def table(self):
        if not self.table_data:
            return ''
        ncol = max(len(r) for r in self.table_data)
        rows = [r + [''] * (ncol - len(r)) for r in self.table_data]
        widths = [max(len(r[c]) for r in rows) for c in range(ncol)]
        sep = '+' + '+'.join('-' * (w + 2) for w in widths) + '+'
        fmt = lambda r: '|' + '|'.join(' %s ' % r[c].ljust(widths[c]) for c in range(ncol)) + '|'
        out = [sep if not self.title else '+' + self.title.center(len(sep) - 2, '-') + '+', fmt(rows[0])]
        if self.inner_heading_row_border and len(rows) > 1:
            out.append(sep)
        out += [fmt(r) for r in rows[1:]]
        out.append(sep)
        return '\n'.join(out)


SingleTable = DoubleTable = AsciiTable
