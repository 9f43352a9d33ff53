"""tests/golden/ref: the golden vectors the reference itself holds beside test.raw (VERDICT r03 "missing" 1).

Copied as DATA from /root/reference (never sources):
  test/8580.wav, test/8580.rec, test/test (-> 8580.mlf), test/lsit.txt   the reference's list-mode + MLF fixture
                                                                          (HU weights through test/PHN_ES)
  es.wav, es.rec                                                          a 19-s utterance through the HU system
  test/PHN_ES/config, test/PHN_ES/dicts/phonemes                          the two files in which test/PHN_ES differs
                                                                          from PHN_HU_SPDAT_LCRC_N1500 (its weights,
                                                                          norms and windows are byte-identical)
and, generated here by the reference CLI built into oracle/_ref (make -C oracle ref):
  8580.mel, 8580.lop        `phnrec_ref -t par` / `-t post` on 8580.wav (the 44-byte RIFF header read as 22 samples,
                            srec.cpp:1384-1422) -- what the host front-end must reproduce bit for bit.
Run in the container that has /root/reference:  python tools/make_golden_ref.py
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "ref")


def main():
    cli = os.path.join(ROOT, "oracle", "_ref", "phnrec_ref")
    if not os.path.exists(cli):
        sys.exit("build the reference first: make -C oracle ref")
    os.makedirs(os.path.join(OUT, "PHN_ES", "dicts"), exist_ok=True)
    for src, dst in (("test/8580.wav", "8580.wav"), ("test/8580.rec", "8580.rec"), ("test/test", "8580.mlf"),
                     ("test/lsit.txt", "lsit.txt"), ("es.wav", "es.wav"), ("es.rec", "es.rec"),
                     ("test/PHN_ES/config", "PHN_ES/config"), ("test/PHN_ES/dicts/phonemes", "PHN_ES/dicts/phonemes")):
        shutil.copyfile(os.path.join(REF, src), os.path.join(OUT, dst))
    es = os.path.join(REF, "test", "PHN_ES")
    for kind, suffix in (("par", "mel"), ("post", "lop")):
        subprocess.check_call([cli, "-c", es, "-i", os.path.join(OUT, "8580.wav"), "-t", kind,
                               "-o", os.path.join(OUT, "8580." + suffix)], stdout=subprocess.DEVNULL)
    print("tests/golden/ref written")


if __name__ == "__main__":
    main()
