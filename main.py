"""Same entry point as the reference's main.py:8-24: ``main.py --mode {train,eval,infer} --model vpho_net ...``."""
from vpho_amd.configs.args import cfg
from vpho_amd.trainer import Trainer


def main():
    trainer = Trainer(cfg)
    if cfg.mode == 'train':
        trainer.run()
    elif cfg.mode == 'eval':
        trainer.eval()
    elif cfg.mode == 'infer':
        trainer.infer()
    else:
        raise ValueError(f'Unknown mode: {cfg.mode}')


if __name__ == '__main__':
    main()
