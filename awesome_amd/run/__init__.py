from .functions import (combine_object_masks, evaluate_dataset, get_result, save_packed_mask_png, save_result_mask,  # noqa: F401
                        split_model_result, write_png_gray)
