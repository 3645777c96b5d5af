"""ViTTA on TANet-R50 / UCF101-C, all 12 corruptions (entry point of the reference kept by name).

Set --model_path, --video_data_dir, --spatiotemp_{mean,var}_clean_file, --val_vid_list ('{}' = corruption)
and --result_dir ('{}_{}/tta_{}') on the command line or edit the Namespace below."""
from vitta_amd.scripts import run_over_corruptions, tanet_ucf101_args

if __name__ == "__main__":
    args = tanet_ucf101_args()
    run_over_corruptions(args)
