"""The reference's own Viterbi golden tests (reference Makefile:116-121,146-186) as data.

Each case: (machine json, reads fasta, CLI error-model flags, golden decoded-bits file, fp64 loglike).
The loglike column is the value recorded in SURVEY.md section 8(c) from the reference run by the
surveyor; the decoded strings are the reference's committed goldens data/hello.{exact,padded}.bits.
"""
NOERRS = dict(sub=0., dup=0., del_open=0., global_=True)          # Makefile:116-120
ONLYDUPS = dict(sub=0., del_open=0., global_=True)                # Makefile:121
DEFAULT = dict()

VITERBI_GOLDENS = [
    # testviterbi, Makefile:146-148
    ("l4c4.json", "hello.fa", NOERRS, "hello.padded.bits", 6.9314718055993767),
    ("l4c4.json", "hello.dup.fa", ONLYDUPS, "hello.padded.bits", 0.9565217636169463),
    # testcompose, Makefile:154
    ("mr2l4c4.json", "hello.mr2.fa", NOERRS, "hello.exact.bits", 13.862943611198832),
    # testham, Makefile:169-171
    ("h74l4c4.json", "hello.h74.fa", NOERRS, "hello.exact.bits", 41.588830833596653),
    ("h74l4c4.json", "hello.h74.fa", DEFAULT, "hello.exact.bits", 40.721062459856107),
    ("h74l4c4.json", "hello.h74.sub.fa", DEFAULT, "hello.exact.bits", 36.030632429917176),
    # testsync, Makefile:177-178
    ("s16mr2l4c4.json", "hello.s16mr2.fa", NOERRS, "hello.exact.bits", 38.816242111356921),
    ("s16mr2l4c4.json", "hello.s16mr2.fa", DEFAULT, "hello.exact.bits", 37.972578414664724),
    # testsyncham, Makefile:184-186
    ("s16h74l4c4.json", "hello.s16h74.fa", NOERRS, "hello.exact.bits", 63.769540611514891),
    ("s16h74l4c4.json", "hello.s16h74.fa", DEFAULT, "hello.exact.bits", 62.708934821387643),
    ("s16h74l4c4.json", "hello.s16h74.del.fa", DEFAULT, "hello.exact.bits", 54.416887183956284),
]

# machine -> (states, usable emit edges, usable null edges, input alphabet); SURVEY.md section 8 table
MACHINE_STATS = {
    "l4c4.json": (384, 686, 27, "$01AB^"),
    "mr2l4c4.json": (1382, 1381, 456, "$01AB^"),
    "h74l4c4.json": (5242, 5311, 1698, "$01AB^"),
    "s16mr2l4c4.json": (8313, 5953, 4718, "$01^"),
    "s16h74l4c4.json": (12361, 11706, 4160, "$01^"),
}
