"""Minimal FASTA reader for the fixtures of the path's callers.

Mirrors the behaviour of /root/reference/fasta/fasta.go:30-58 (Read) as far as the align callers
need it: records start with '>', sequence lines are concatenated, bases go through dna.StringToBases
(case preserved -- callers upper-case explicitly, cmd/cigarToBed/cigarToBed.go:69-70).
"""
from . import dna


class Fasta:
    __slots__ = ("Name", "Seq")

    def __init__(self, Name, Seq):
        self.Name = Name
        self.Seq = Seq


def Read(filename):
    records, name, chunks = [], None, []
    with open(filename) as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if not line or line.startswith("#"):  # fileio.EasyNextRealLine skips comment lines
                continue
            if line.startswith(">"):
                if name is not None:
                    records.append(Fasta(name, dna.StringToBases("".join(chunks))))
                name, chunks = line[1:], []
            else:
                chunks.append(line)
    if name is not None:
        records.append(Fasta(name, dna.StringToBases("".join(chunks))))
    return records


def ToMap(records):
    m = {}
    for r in records:
        if r.Name in m:
            raise RuntimeError("%s used for multiple fasta records. record names must be unique." % r.Name)
        m[r.Name] = r.Seq
    return m


def WriteFasta(fh, rec, lineLength=50):
    """fasta.WriteFasta (/root/reference/fasta/fasta.go:163-177): name line, then lines of lineLength bases."""
    fh.write(">%s\n" % rec.Name)
    s = dna.BasesToString(rec.Seq)
    for i in range(0, len(s), lineLength):
        fh.write(s[i:i + lineLength] + "\n")


def Write(filename, records):
    """fasta.Write (fasta.go:147-152): line length 50."""
    with open(filename, "w") as fh:
        for rec in records:
            WriteFasta(fh, rec, 50)


def ToUpper(fa):
    dna.AllToUpper(fa.Seq)


def AllAreEqualIgnoreOrder(alpha, beta):
    """fasta.AllAreEqualIgnoreOrder (/root/reference/fasta/fasta.go): same records (name + sequence) in any order."""
    if len(alpha) != len(beta):
        return False
    key = lambda f: (f.Name, bytes(bytearray(f.Seq)))
    return sorted(map(key, alpha)) == sorted(map(key, beta))
