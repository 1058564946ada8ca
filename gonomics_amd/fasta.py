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
            if not line:
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
    return {r.Name: r.Seq for r in records}


def ToUpper(fa):
    dna.AllToUpper(fa.Seq)


def AllAreEqualIgnoreOrder(alpha, beta):
    """fasta.AllAreEqualIgnoreOrder (/root/reference/fasta/fasta.go): same records (name + sequence) in any order."""
    if len(alpha) != len(beta):
        return False
    key = lambda f: (f.Name, bytes(bytearray(f.Seq)))
    return sorted(map(key, alpha)) == sorted(map(key, beta))
