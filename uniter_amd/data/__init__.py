"""Host-side input pipeline: record stores and codecs of the reference's databases, the dataset
classes and batch builders of the four training configurations, token-bucket batching, the multi-task loader and the
prefetching loader (H2D + bf16 cast on a side HIP stream).  The exported names are those of the reference's `data` package
(data/__init__.py) for the tasks SURVEY section 8 keeps in scope."""
from .loader import MetaLoader, PrefetchLoader, move_to_cuda, record_cuda_stream  # noqa: F401
from .sampler import TokenBucketSampler  # noqa: F401
from .collate import get_gather_index, pad_tensors, sequence_lengths  # noqa: F401
from .data import (TxtTokLmdb, TxtLmdb, DetectFeatLmdb, ImageLmdbGroup, ConcatDatasetWithLens,  # noqa: F401
                   DetectFeatTxtTokDataset, compute_num_bb, get_ids_and_lens, open_lmdb)
from .store import FeaturePack, LmdbStore, PackStore, PackWriter, convert_store, open_store  # noqa: F401
from .tasks import (MlmDataset, mlm_collate, MrfrDataset, MrcDataset, mrfr_collate, mrc_collate,  # noqa: F401
                    TokenBucketSamplerForItm, ItmDataset, itm_collate, itm_ot_collate,
                    ItmRankDataset, ItmValDataset, ItmEvalDataset, itm_rank_collate, itm_val_collate, itm_eval_collate,
                    ItmRankDatasetHardNegFromText, ItmRankDatasetHardNegFromImage, itm_rank_hn_collate,
                    VeDataset, VeEvalDataset, ve_collate, ve_eval_collate,
                    Nlvr2PairedDataset, Nlvr2PairedEvalDataset, Nlvr2TripletDataset, Nlvr2TripletEvalDataset,
                    nlvr2_paired_collate, nlvr2_paired_eval_collate, nlvr2_triplet_collate, nlvr2_triplet_eval_collate,
                    VqaDataset, VqaEvalDataset, vqa_collate, vqa_eval_collate, joint_batch, random_word)
