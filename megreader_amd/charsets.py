"""Minimal charset objects (only what the decoders need: len(), blank, unknown).

Mirrors reference concern/charsets.py:10-27,101-110: EnglishCharset = 36 alphanumerics with blank (id 0) and
unknown (id 1) inserted in front => 38 classes.  The reference's own Charset objects (constructed from YAML)
are accepted unchanged by the decoders; this class is the default when none is passed.
"""
import string


class EnglishCharset(object):
    blank = 0
    unknown = 1
    case_sensitive = False

    def __init__(self):
        corpus = sorted(set(string.digits + string.ascii_uppercase))
        self._charset = [None, None] + corpus  # reference quirk Q6: blank_char/unknown_char default to None

    def __len__(self):
        return len(self._charset)

    def __getitem__(self, index):
        return self._charset[index]

    def is_empty(self, index):
        return index == self.blank or index == self.unknown

    def index(self, x):
        """concern/charsets.py:37-41"""
        target = x if self.case_sensitive else x.upper()
        try:
            return self._charset.index(target)
        except ValueError:
            return self.unknown

    def label_to_string(self, label):
        ignore = (self.unknown, self.blank)
        return "".join(self._charset[int(i)] for i in label if int(i) not in ignore)


DefaultCharset = EnglishCharset
