# Default solver constants (same names and values the reference imports from src/option.py:2-10)
default_hparas = {
    'GRAD_CLIP': 5.0,
    'PROGRESS_STEP': 100,
    'DEV_STEP_RATIO': 1.2,
    'DEV_N_EXAMPLE': 4,
    'TB_FLUSH_FREQ': 180,
}
