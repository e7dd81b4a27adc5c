// MNIST CNN training on MI355X -- counterpart of the reference's examples/train_mnist_cnn.rs
// (model of lines 35-100, Adam(1e-2, wd 1e-4), batch 256, images reshaped to [B,1,28,28]).
// Faithful by default: the reference cuts the tape at im2col / transpose_4d, so only the last conv's
// bias and the classifier train (quirk Q2); --full-backward trains every conv weight (extension).
#include <chrono>

#include "common.h"

using namespace taper;

int main(int argc, char **argv) {
    bool full = false;
    int kept = 1;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--full-backward")) full = true;
        else argv[kept++] = argv[i];
    }
    ex::Args args = ex::parse(kept, argv);
    if (args.epochs == 0) args.epochs = 50;  // train_mnist_cnn.rs:115
    try {
        printf("MNIST CNN Training\n\nLoading MNIST dataset...\n");
        MNISTDataset train_ds = ex::load(args, true), test_ds = ex::load(args, false);
        printf("Training set: %zu samples\nTest set: %zu samples\n\n", train_ds.len(), test_ds.len());
        DataLoader train_loader(train_ds, args.batch_size, true), test_loader(test_ds, args.batch_size, false);
        set_full_backward(full);

        printf("Building optimized CNN model...\n");
        auto conv = [](size_t ci, size_t co, uint64_t seed) {
            return std::make_shared<Conv2dReLU>(ci, co, std::make_pair(3, 3), std::make_pair(1, 1), std::make_pair(1, 1), true, seed);
        };
        auto pool = [] { return std::make_shared<MaxPool2d>(std::make_pair(2, 2), std::make_pair(2, 2), std::make_pair(0, 0)); };
        auto model = std::make_shared<Sequential>(std::vector<std::shared_ptr<Module>>{
            conv(1, 32, 1), conv(32, 32, 2), pool(),                       // 28x28 -> 14x14      (train_mnist_cnn.rs:37-58)
            conv(32, 64, 3), conv(64, 64, 4), pool(),                      // 14x14 -> 7x7        (:60-81)
            conv(64, 128, 5),                                              //                     (:83-92)
            std::make_shared<AdaptiveAvgPool2d>(std::make_pair(1, 1)), std::make_shared<Flatten>(1),   // (:94-95)
            std::make_shared<Linear>(128, 128, true, 6), std::make_shared<ReLU>(),                     // (:97-101)
            std::make_shared<Linear>(128, 64, true, 7), std::make_shared<ReLU>(), std::make_shared<Linear>(64, 10, true, 8)});
        size_t n_params = 0;
        for (const Tensor &p : model->parameters()) n_params += p.len();
        printf("Total parameters: %zu\n", n_params);
        float lr = 0.01f;
        auto optimizer = std::make_shared<Adam>(model->parameters(), lr, 0.9f, 0.999f, 1e-8f, 0.0001f);  // train_mnist_cnn.rs:108-109
        Trainer trainer(model, optimizer);
        trainer.sample_shape = {1, 28, 28};                                                              // train_mnist_cnn.rs:161-162
        printf("\nTraining Configuration:\n   Batch size: %zu\n   Learning rate: %g\n   Epochs: %zu\n   Conv gradients: %s\n\n", args.batch_size, lr,
               args.epochs, full ? "full backward (extension)" : "faithful (conv weights frozen, as the reference)");

        for (size_t epoch = 1; epoch <= args.epochs; ++epoch) {
            const auto t0 = std::chrono::steady_clock::now();
            if (epoch % 5 == 0 && epoch >= 5) {  // train_mnist_cnn.rs:132-137; lr lives on the device, the captured graph reads it
                lr *= 0.8f;
                printf("   Reducing learning rate to %.6f\n", lr);
                optimizer->set_lr(lr);
            }
            const EpochResult tr = args.eager ? trainer.train_epoch(train_loader) : trainer.train_epoch_graph(train_loader);
            const EpochResult va = trainer.evaluate(test_loader);
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("Epoch %zu/%zu:\n   Train Loss: %.4f | Train Acc: %.2f%%\n   Val Loss: %.4f   | Val Acc: %.2f%%\n   Time: %.2fs (%.0f samples/s)\n\n",
                   epoch, args.epochs, tr.avg_loss, tr.accuracy * 100.f, va.avg_loss, va.accuracy * 100.f, secs, tr.total_samples / secs);
            if (va.accuracy > 0.995f) {  // train_mnist_cnn.rs:261-267
                printf("Reached %.2f%% validation accuracy! Stopping early.\n", va.accuracy * 100.f);
                break;
            }
        }
        printf("Training Complete!\n");
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
