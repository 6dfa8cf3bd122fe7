"""rec_pangu_amd — MI355X-native implementation of rec_pangu's ranking forward/backward hot path.

Public surface mirrors the reference for that path only:
    rec_pangu_amd.models.ranking.{DeepFM, xDeepFM, DCN, AutoInt, FM}
    rec_pangu_amd.models.multi_task.MMOE
    rec_pangu_amd.trainer.RankTrainer, rec_pangu_amd.benchmark_trainer.BenchmarkTrainer
    rec_pangu_amd.dataset.get_dataloader
Kernels: rec_pangu_amd/csrc/*.hip behind the C ABI in include/rec_pangu_hip.h, bound by rec_pangu_amd.hip.
"""
__version__ = "0.1.0"
