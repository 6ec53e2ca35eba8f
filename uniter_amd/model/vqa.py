"""UNITER for VQA fine-tuning.  Reference: model/vqa.py:17-52 (pooler -> Linear(H,2H)+GELU+LN(2H)+Linear(2H,A),
un-reduced BCE-with-logits; train_vqa.py:188 later takes mean() * n_answers)."""
from collections import defaultdict

from torch import nn
from torch.nn import functional as F

from .layer import GELU, BertLayerNorm as LayerNorm
from .model import UniterModel, UniterPreTrainedModel


class UniterForVisualQuestionAnswering(UniterPreTrainedModel):
    def __init__(self, config, img_dim, num_answer):
        super().__init__(config)
        self.uniter = UniterModel(config, img_dim)
        wide = config.hidden_size * 2
        self.vqa_output = nn.Sequential(nn.Linear(config.hidden_size, wide), GELU(), LayerNorm(wide, eps=1e-12),
                                        nn.Linear(wide, num_answer))
        self.apply(self.init_weights)

    def forward(self, batch, compute_loss=True):
        batch = defaultdict(lambda: None, batch)
        self.uniter.seq_lens_hint = batch['seq_lens']          # optional host-side lengths (packed execution, no sync)
        sequence_output = self.uniter(batch['input_ids'], batch['position_ids'], batch['img_feat'],
                                      batch['img_pos_feat'], batch['attn_masks'], batch['gather_index'],
                                      output_all_encoded_layers=False)
        answer_scores = self.vqa_output(self.uniter.pooler(sequence_output))
        if not compute_loss:
            return answer_scores
        return F.binary_cross_entropy_with_logits(answer_scores.float(), batch['targets'].float(), reduction='none')
