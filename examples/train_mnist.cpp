// MNIST MLP training on MI355X -- counterpart of the reference's examples/train_mnist.rs
// (same model 784-128-64-10, Adam(1e-3, wd 1e-4), batch 256, 10 epochs, early stop at 98 %).
//   --eager  runs the reference's loop literally (examples/train_mnist.rs:89-121: Tape::reset,
//            forward, cross_entropy_loss, accuracy, backward, step, zero_grad, loss read-back per batch);
//   default  the same step captured once into a hipGraph and replayed (Trainer::train_epoch_graph).
#include <chrono>

#include "common.h"

using namespace taper;

int main(int argc, char **argv) {
    ex::Args args = ex::parse(argc, argv);
    if (args.epochs == 0) args.epochs = 10;  // train_mnist.rs:62
    try {
        printf("MNIST Neural Network Training\n\nLoading MNIST dataset...\n");
        MNISTDataset train_ds = ex::load(args, true), test_ds = ex::load(args, false);
        printf("Training set: %zu samples\nTest set: %zu samples\n\n", train_ds.len(), test_ds.len());
        DataLoader train_loader(train_ds, args.batch_size, true), test_loader(test_ds, args.batch_size, false);

        printf("Building model...\n");
        auto model = std::make_shared<Sequential>(std::vector<std::shared_ptr<Module>>{
            std::make_shared<Linear>(784, 128, true, 1), std::make_shared<ReLU>(),   // train_mnist.rs:35-41
            std::make_shared<Linear>(128, 64, true, 2), std::make_shared<ReLU>(),
            std::make_shared<Linear>(64, 10, true, 3)});
        size_t n_params = 0;
        for (const Tensor &p : model->parameters()) n_params += p.len();
        printf("Total parameters: %zu\n", n_params);
        const float lr = 0.001f;
        auto optimizer = std::make_shared<Adam>(model->parameters(), lr, 0.9f, 0.999f, 1e-8f, 0.0001f);  // train_mnist.rs:50-51
        Trainer trainer(model, optimizer);
        printf("\nTraining Configuration:\n   Batch size: %zu\n   Learning rate: %g\n   Epochs: %zu\n   Step: %s\n\n", args.batch_size, lr,
               args.epochs, args.eager ? "eager (per-op launches, loss read back every batch)" : "hipGraph replay");

        for (size_t epoch = 1; epoch <= args.epochs; ++epoch) {
            const auto t0 = std::chrono::steady_clock::now();
            printf("Epoch %zu/%zu\n", epoch, args.epochs);
            EpochResult tr;
            if (args.eager) {
                train_loader.reset();
                Tensor images, labels;
                float loss_sum = 0.f;
                while (train_loader.next(&images, &labels)) {
                    float loss, acc;
                    trainer.train_step(images, labels, &loss, &acc);
                    tr.total_correct += (size_t)(acc * (float)labels.len());   // truncation as train_mnist.rs:109 (Q13)
                    tr.total_samples += labels.len();
                    loss_sum += loss;
                    if (++tr.num_batches % 100 == 0)
                        printf("\r   Batch [%zu/%zu] Loss: %.4f, Acc: %.2f%%", tr.num_batches, train_loader.num_batches(), loss,
                               100.0 * tr.total_correct / tr.total_samples), fflush(stdout);
                }
                tr.avg_loss = loss_sum / (float)tr.num_batches;
                tr.accuracy = (float)tr.total_correct / (float)tr.total_samples;
                printf("\n");
            } else {
                tr = trainer.train_epoch_graph(train_loader);
            }
            const EpochResult va = trainer.evaluate(test_loader);
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("Epoch %zu complete:\n   Train Loss: %.4f | Train Acc: %.2f%%\n   Val Loss: %.4f   | Val Acc: %.2f%%\n   Time: %.2fs (%.0f samples/s)\n\n",
                   epoch, tr.avg_loss, tr.accuracy * 100.f, va.avg_loss, va.accuracy * 100.f, secs, tr.total_samples / secs);
            if (va.accuracy > 0.98f) {  // train_mnist.rs:188-194
                printf("Reached %.2f%% validation accuracy! Stopping early.\n", va.accuracy * 100.f);
                break;
            }
        }
        printf("\nTraining Complete!\n");
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
