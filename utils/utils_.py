"""utils/utils_.py of the reference -> vitta_amd.utils_ (the helpers the TTA path touches)."""
from vitta_amd.utils_ import (AverageMeter, AverageMeterTensor, MovingAverageTensor, accuracy,  # noqa: F401
                              get_writer_to_all_result, make_dir, model_analysis, path_logger)
