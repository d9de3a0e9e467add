"""Host-side mirror of the reference's sparse::SparseMatrix, just enough to build a parity
check matrix in Python and hand it to the decoder boundary as alist text (what a Rust shim
does with `h.alist()`).  Semantics follow /root/reference/src/sparse.rs:114-119 (insert
de-duplicates and appends), :250-299 (alist writer) and :352-389 (alist reader: column
section only, zero padding ignored)."""


class SparseMatrix:
    def __init__(self, nrows: int, ncols: int):
        self.rows = [[] for _ in range(nrows)]
        self.cols = [[] for _ in range(ncols)]

    def num_rows(self):
        return len(self.rows)

    def num_cols(self):
        return len(self.cols)

    def row_weight(self, r):
        return len(self.rows[r])

    def col_weight(self, c):
        return len(self.cols[c])

    def contains(self, r, c):
        return r in self.cols[c]

    def insert(self, r, c):
        if not self.contains(r, c):
            self.rows[r].append(c)
            self.cols[c].append(r)

    def insert_row(self, r, cols):
        for c in cols:
            self.insert(r, c)

    def insert_col(self, c, rows):
        for r in rows:
            self.insert(r, c)

    def iter_row(self, r):
        return iter(self.rows[r])

    def iter_col(self, c):
        return iter(self.cols[c])

    def _write(self, padding):
        out = [f"{self.num_cols()} {self.num_rows()}"]
        dirs = (self.cols, self.rows)
        maxlen = [max((len(x) for x in d), default=0) for d in dirs]
        out.append(f"{maxlen[0]} {maxlen[1]}")
        for d in dirs:
            out.append(" ".join(str(len(x)) for x in d))
        for d, ml in zip(dirs, maxlen):
            for x in d:
                toks = [str(v + 1) for v in sorted(x)]
                if padding:
                    if not toks:
                        toks = ["0"]
                    toks += ["0"] * (ml - max(len(x), 1))
                out.append(" ".join(toks))
        return "\n".join(out) + "\n"

    def alist(self):
        return self._write(True)

    def alist_no_padding(self):
        return self._write(False)

    @staticmethod
    def from_alist(text: str) -> "SparseMatrix":
        lines = text.split("\n")
        if not lines:
            raise ValueError("alist first line not found")
        first = lines[0].split()
        if len(first) < 2:
            raise ValueError("alist first line does not contain enough elements")
        try:
            ncols = int(first[0])
        except ValueError:
            raise ValueError("ncols is not a number")
        try:
            nrows = int(first[1])
        except ValueError:
            raise ValueError("nrows is not a number")
        h = SparseMatrix(nrows, ncols)
        for c in range(ncols):
            if 4 + c >= len(lines):
                raise ValueError("alist does not contain expected number of lines")
            for tok in lines[4 + c].split():
                try:
                    r = int(tok)
                except ValueError:
                    raise ValueError("row value is not a number")
                if r != 0:
                    h.insert(r - 1, c)
        return h
