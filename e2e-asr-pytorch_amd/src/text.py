"""Character tokenizer with the reference's conventions (src/text.py:45-93): <pad>=0, <eos>=1, <unk>=2,
`encode` appends <eos>, `decode` stops at <eos>, skips <pad> and (for CTC) collapses repeats."""


class CharacterTextEncoder(object):
    pad_idx, eos_idx, unk_idx = 0, 1, 2
    token_type = 'character'

    def __init__(self, vocab_list):
        self._vocab_list = ['<pad>', '<eos>', '<unk>'] + list(vocab_list)
        self._vocab2idx = {v: i for i, v in enumerate(self._vocab_list)}

    @classmethod
    def load_from_file(cls, vocab_file):
        with open(vocab_file, 'r', encoding='UTF-8') as f:
            return cls([line.strip('\r\n') for line in f])

    @property
    def vocab_size(self):
        return len(self._vocab_list)

    def encode(self, s):
        s = s.strip('\r\n ')
        return [self._vocab2idx.get(v, self.unk_idx) for v in s] + [self.eos_idx]

    def decode(self, idxs, ignore_repeat=False):
        out = []
        for t, idx in enumerate(idxs):
            if idx == self.eos_idx:
                break
            if idx == self.pad_idx or (ignore_repeat and t > 0 and idx == idxs[t - 1]):
                continue
            out.append(self._vocab_list[idx])
        return ''.join(out)


def load_text_encoder(mode, vocab_file):
    if mode != 'character':
        raise NotImplementedError('only the character tokenizer is part of the HIP build (got %s)' % mode)
    return CharacterTextEncoder.load_from_file(vocab_file)
