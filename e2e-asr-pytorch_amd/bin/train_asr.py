"""Training Solver with the reference's surface (bin/train_asr.py:17-394): `load_data`, `set_model`, `exec`,
`fetch_data`, `validate`.  Every tensor op of the step runs in libasr_hip.so (src/asr.py, src/util.py)."""
import torch

from src import hipabi as H
from src.asr import ASR
from src.data import load_dataset
from src.optim import Optimizer
from src.solver import BaseSolver
from src.util import human_format, cal_er, CTCLoss, CrossEntropyLoss, LabelSmoothingLoss


class Solver(BaseSolver):
    def __init__(self, config, paras, mode):
        super().__init__(config, paras, mode)
        self.curriculum = self.config['hparas'].get('curriculum', 0)
        self.val_mode = self.config['hparas'].get('val_mode', 'wer').lower()
        self.WER = 'per' if self.val_mode == 'per' else 'wer'

    def fetch_data(self, data, train=False):
        """Batch to the device; a WAVEFORM batch (B,N) goes through the GPU front-end here - fbank, delta stack and, in
        training, SpecAugment - which the reference runs per utterance on the CPU inside its DataLoader workers
        (src/audio.py:453-486, src/collect_batch.py:32)."""
        _, feat, feat_len, txt = data
        feat, feat_len, txt = feat.to(self.device), feat_len.to(self.device), txt.to(self.device)
        if feat.dim() == 2:
            tf = (self.tr_set if train else self.dv_set).audio_transform
            with torch.no_grad():
                feat, feat_len = tf(feat, feat_len)
        return feat, feat_len, txt, torch.sum(txt != 0, dim=-1)

    def load_data(self):
        audio = dict(self.config['data']['audio'])
        audio.setdefault('time_aug', False)          # omitted by config/librispeech_asr.yaml
        self.tr_set, self.dv_set, self.feat_dim, self.vocab_size, self.tokenizer, msg = load_dataset(
            self.paras.njobs, self.paras.gpu, self.paras.pin_memory, self.curriculum > 0,
            self.config['data']['corpus'], audio, self.config['data']['text'], rank=self.rank, world=self.world)
        self.verbose(msg)
        for ld in (self.tr_set, self.dv_set):
            if getattr(ld, 'audio_transform', None) is not None:
                ld.audio_transform = ld.audio_transform.to(self.device)
        self.dv_names = self.config['data']['corpus']['dev_split'][0]
        self.best_wer = {'att': {self.dv_names: 3.0}, 'ctc': {self.dv_names: 3.0}}

    def set_model(self):
        hip = self.config.get('hip', {})
        batch_size = self.config['data']['corpus']['batch_size'] // 2
        self.model = ASR(self.feat_dim, self.vocab_size, batch_size, prec=hip.get('prec', 'bf16'),
                         seed=self.paras.seed + 7919 * self.rank, **self.config['model']).to(self.device)
        self.verbose(self.model.create_msg())
        hp = dict(self.config['hparas'])
        if hp.get('label_smoothing', False):        # default False when the YAML omits it
            self.seq_loss = LabelSmoothingLoss(31, 0.1)
        else:
            self.seq_loss = CrossEntropyLoss(ignore_index=0)
        self.ctc_loss = CTCLoss(blank=0, zero_infinity=False)
        self.optimizer = Optimizer(self.model.parameters(), **hp)
        self.lr_scheduler = self.optimizer.lr_scheduler
        self.verbose(self.optimizer.create_msg())
        from src import dist as D_
        if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            self.dp = self.model.attach_data_parallel()
            self.dp.broadcast_params(0)
        self.load_ckpt()

    def exec(self):
        # single-process runs overlap the parameter gradients with the BPTT on CU-masked streams, which serialise against
        # the legacy default stream (src/hipabi.work_stream): the whole loop runs on a non-default stream then
        if (self.dp is None or H.overlap_dp_enabled()) and H.overlap_enabled() and torch.cuda.is_available():
            ws = H.work_stream()
            ws.wait_stream(torch.cuda.current_stream())     # model init / flatten / broadcast / load_ckpt ran on the default stream
            with torch.cuda.stream(ws):
                self._exec()
            torch.cuda.current_stream().wait_stream(H.work_stream())
        else:
            self._exec()

    def _exec(self):
        self.verbose('Total training steps {}.'.format(human_format(self.max_step)))
        self.timer.set()
        while self.step < self.max_step:
            for data in self.tr_set:
                tf_rate = self.optimizer.pre_step(self.step)
                total_loss, ctc_loss, att_loss = 0, None, None
                feat, feat_len, txt, txt_len = self.fetch_data(data, train=True)
                self.timer.cnt('rd')
                L = int(txt.shape[1])      # = max(txt_len) for a padded batch, without a device sync
                ctc_output, encode_len, att_output, att_align, _ = self.model(feat, feat_len, L, tf_rate=tf_rate, teacher=txt, ctc_async=True)
                if ctc_output is not None:
                    if getattr(ctc_output, '_asr_side', False):      # CTC branch on the side stream (src/asr.py, src/step.py)
                        with H.side_branch(False, txt, txt_len):
                            ctc_loss = self.ctc_loss(ctc_output.transpose(0, 1), txt, encode_len, txt_len)
                        H.join_branch(ctc_loss, ctc_output)
                    else:
                        ctc_loss = self.ctc_loss(ctc_output.transpose(0, 1), txt, encode_len, txt_len)
                    total_loss = total_loss + ctc_loss * self.model.ctc_weight
                if att_output is not None:
                    b, t, _ = att_output.shape
                    att_loss = self.seq_loss(att_output.view(b * t, -1), txt[:, :t].reshape(-1))
                    w = 1 - self.model.ctc_weight
                    if self.dp is not None and self.dp.world > 1:
                        w = w * self.dp.ce_weight(txt_len.sum())
                    total_loss = total_loss + att_loss * w
                self.timer.cnt('fw')
                grad_norm = self.backward(total_loss)
                self.step += 1
                if (self.step == 1) or (self.step % self.PROGRESS_STEP == 0):
                    H.raise_if_aborted()      # the host synchronises here anyway (loss.item()): surface a refused step
                    self.progress('Tr stat | Loss - {:.2f} | Grad. Norm - {:.2f} | {}'.format(
                        total_loss.item(), grad_norm.item(), self.timer.show()))
                    if att_output is not None:
                        self.write_log('loss', {'tr_att': att_loss})
                        self.write_log(self.WER, {'tr_att': cal_er(self.tokenizer, att_output, txt)})
                    if ctc_output is not None:
                        self.write_log('loss', {'tr_ctc': ctc_loss})
                        self.write_log(self.WER, {'tr_ctc': cal_er(self.tokenizer, ctc_output, txt, ctc=True)})
                if (self.step == 1) or (self.step % self.valid_step == 0):
                    self.validate(self.dv_set, self.dv_names)
                if self.lr_scheduler is None and self.step > 99999 and self.step % 2000 == 0:
                    for g in self.optimizer.opt.param_groups:      # hand-rolled decay, bin/train_asr.py:292-303
                        g['lr'] = g['lr'] * 0.85
                self.timer.set()
                if self.step >= self.max_step:
                    break
        self.log.close()
        print('[INFO] Finished training after', human_format(self.max_step), 'steps.')

    def validate(self, _dv_set, _name):
        self.model.eval()
        dev_er = {'att': [], 'ctc': []}
        for i, data in enumerate(_dv_set):
            self.progress('Valid step - {}/{}'.format(i + 1, len(_dv_set)))
            feat, feat_len, txt, txt_len = self.fetch_data(data)
            with torch.no_grad():
                ctc_output, encode_len, att_output, att_align, _ = self.model(
                    feat, feat_len, int(txt.shape[1] * self.DEV_STEP_RATIO))
            if att_output is not None:
                dev_er['att'].append(cal_er(self.tokenizer, att_output, txt, mode=self.val_mode))
            if ctc_output is not None:
                dev_er['ctc'].append(cal_er(self.tokenizer, ctc_output, txt, mode=self.val_mode, ctc=True))
        H.raise_if_aborted()
        for task in [k for k, v in dev_er.items() if len(v) > 0]:
            er = sum(dev_er[task]) / len(dev_er[task])
            if er < self.best_wer[task][_name]:
                self.best_wer[task][_name] = er
                self.save_checkpoint('best_{}_{}.pth'.format(task, _name), self.val_mode, er, _name)
            if self.step >= self.max_step:
                self.save_checkpoint('last_{}_{}.pth'.format(task, _name), self.val_mode, er, _name)
            self.write_log(self.WER, {'dv_' + task + '_' + _name.lower(): er})
        self.model.train()
